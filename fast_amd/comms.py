"""Link metrics over the power vector: the reference's fast/comms.py:171-262 surface
(fade_prob, fade_dur, ber_ook, sep_qam, ber_qam, Q), evaluated on the GPU through
fastmc_link_metrics.  Same names, arguments and NaN conventions as the reference.

Each function takes the sample vector the reference takes (uploaded for the call).  Pass a
`fast_amd.Fast` object instead of an array to reduce the last run's results where they already
are, on the device, without moving the vector (order-independent metrics only: fade_dur needs
the FastResult ordering and always takes the vector).  With a `Fast` object every function works on
the power RELATIVE to the diffraction limit (`FastResult._r`, i.e. 10^(dB_rel/10)): thresholds are in
that unit for fade_prob, fade_counts and fade_dur alike; pass `sim.result.power` (an array, watts)
to work in watts.

The Monte-Carlo symbol simulator (Modulator / FastFSOC, comms.py:13-168) and the GMI tools are
outside the hot-path scope and are not provided.
"""
import numpy

from . import _lib


def _metrics(queries, samples):
    from .fast import Fast
    if isinstance(samples, Fast):
        return _lib.link_metrics(queries, handle=samples._handle)
    dev = _lib.default_device()
    return _lib.link_metrics(queries, samples=numpy.abs(samples) ** 2 if numpy.iscomplexobj(samples) else samples, device=dev)


def fade_counts(I, threshold):
    """(n, n_below, n_complete, samples_in_complete): the integers fade_prob and fade_dur are formed
    from.  A fade is complete when it starts after the first sample and ends before the last
    (comms.py:181-186: rising edges of the mask, final segment dropped when it is still fading)."""
    from .fast import Fast
    if isinstance(I, Fast):
        # fade durations need the FastResult ordering: take its vector -- in the SAME unit fade_prob reduces on the
        # device (power relative to the diffraction limit, FastResult._r; |a|^2 for a coherent run), so that one
        # threshold means the same thing in fade_prob(sim, t) and fade_dur(sim, t)
        I = I.result._r
        I = numpy.abs(I) ** 2 if numpy.iscomplexobj(I) else I
    I = numpy.asarray(I)
    n = I.size
    n_below, n_rise, first_clear, last_clear = _lib.link_metrics([(_lib.LM_FADE, threshold, 0.0)], samples=I,
                                                                 device=_lib.default_device())[0]
    n_below, n_rise = int(n_below), int(n_rise)
    lead = int(first_clear)                       # length of a fade already running at sample 0
    trail = n - 1 - int(last_clear)               # length of a fade still running at the end
    if n_rise == 0:                               # no fade starts inside the record
        return n, n_below, 0, 0
    open_end = 1 if trail > 0 else 0
    return n, n_below, n_rise - open_end, n_below - lead - (trail if open_end else 0)


def fade_prob(I, threshold, min_fades=30):
    """comms.py:171-177.  I may be a Fast object (relative powers of the last run, on the device)."""
    from .fast import Fast
    if isinstance(I, Fast):
        n_below = int(_lib.link_metrics([(_lib.LM_FADE, threshold, 0.0)], handle=I._handle)[0, 0])
        n = I._handle.result_stats()["n"]
    else:
        n, n_below, _, _ = fade_counts(I, threshold)
    if n_below < min_fades:
        return numpy.nan
    return n_below / n


def fade_dur(I, threshold, dt=1, min_fades=30):
    """comms.py:180-195: mean duration of the complete fades, NaN with fewer than min_fades of them."""
    _, _, n_complete, total = fade_counts(I, threshold)
    if n_complete < min_fades:
        return numpy.nan
    return total / n_complete * dt


def ber_ook(EbN0, samples=None):
    """comms.py:198-222.  samples=None: no atmosphere (one sample of unit power)."""
    s, _, n, _ = _metrics([(_lib.LM_BER_OOK, EbN0, 0.0)], numpy.ones(1) if samples is None else samples)[0]
    return s / n


def sep_qam(M, EsN0, samples=None):
    """comms.py:225-242."""
    s, _, n, _ = _metrics([(_lib.LM_SEP_QAM, M, EsN0)], numpy.ones(1) if samples is None else samples)[0]
    return s / n


def ber_qam(M, EbN0, samples=None):
    """comms.py:245-255: nearest-neighbour errors with Gray coding."""
    return 1 / numpy.log2(M) * sep_qam(M, 10 * numpy.log10(numpy.log2(M)) + EbN0, samples)


def Q(x):
    """comms.py:258-262, through the same device code: Q(x) = ber_ook at snr = x with no atmosphere.  ONE library call for the
    whole array (the queries of every non-zero element together; Q(-x) = 1 - Q(x), Q(0) = 1/2)."""
    x = numpy.asarray(x, dtype=float)
    flat = numpy.atleast_1d(x).ravel()
    out = numpy.full(flat.shape, 0.5)
    nz = numpy.flatnonzero(flat != 0)
    if nz.size:
        queries = [(_lib.LM_BER_OOK, float(20 * numpy.log10(abs(v))), 0.0) for v in flat[nz]]
        res = _metrics(queries, numpy.ones(1))
        q = numpy.array([r[0] / r[2] for r in res])
        out[nz] = numpy.where(flat[nz] > 0, q, 1.0 - q)
    return out.reshape(x.shape) if x.ndim else float(out[0])
