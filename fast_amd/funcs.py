"""The two screen generators of the reference's fast/funcs.py with their own signatures, on the GPU:

    make_phase_fft(rand, df, fftw, fftw_objs, double)     fast/funcs.py:210-223
    make_phase_subharm(rand, freq, N, dx, double)         fast/funcs.py:225-258

`Fast.run()` never materialises full N x N screens (it keeps the pupil window and reduces it on the
device); these functions do, like the reference's, for callers that want the screens themselves.
The transform follows the reference's FFTW branch (numpy fftshift, unnormalised forward DFT, fftshift)
whatever the `fftw` flag says: the other branch is aotools.fouriertransform.ift2, which is not
available here.  Helpers that are pure bookkeeping (generate_random_coefficients,
l_path) keep the reference's behaviour on the host.
"""
import numpy

from . import _lib, host


def _full_window_handle(N, device, dx=1.0):
    h = _lib.Handle(N, N, "f64", device)
    h.set_pupil(numpy.ones((N, N)), 0, float(dx))     # window = the whole grid; dx places the sub-harmonic modes
    return h


def make_phase_fft(rand, df, fftw=False, fftw_objs=None, double=False, device=None):
    """rand: (B, N, N) complex, already coloured with sqrt(powerspec).  Returns the (B, N, N) real
    screens Re{fftshift(FFT2(fftshift(rand df)))}, or with double=True the (2B, N, N) stack [Re | Im]."""
    rand = numpy.asarray(rand)
    squeeze = rand.ndim == 2
    r = rand[numpy.newaxis] if squeeze else rand
    B, N = r.shape[0], r.shape[-1]
    if r.shape[1:] != (N, N):
        raise ValueError("rand must be (..., N, N)")
    h = _full_window_handle(N, device)
    try:
        h.set_spectrum(numpy.ones((N, N)), float(df))             # amp = sqrt(1) * df
        phs = h.screens_coeffs(numpy.ascontiguousarray(r.real, dtype=float), numpy.ascontiguousarray(r.imag, dtype=float))
    finally:
        h.close()
    if double:
        return phs                                               # numpy.vstack([screens.real, screens.imag])
    out = phs[:B]
    return out[0] if squeeze else out


def make_phase_subharm(rand, freq, N, dx, double=False, device=None):
    """rand: (B, 3, 3, 3) complex, already coloured; freq: an object with .subharm.fx / .fy (3, 3, 3)
    and .subharm.df (3,) like the reference's SpatialFrequencies (fast_amd.Fast(...).freq works).
    Sum of the 27 low-frequency modes on the N x N grid, per-screen complex mean removed."""
    rand = numpy.asarray(rand)
    B = rand.shape[0]
    sh = freq.subharm
    h = _full_window_handle(N, device, dx)                        # the modes are evaluated at arange(-D/2, D/2, dx)
    try:
        h.set_spectrum(numpy.zeros((N, N)), 1.0)                  # no FFT contribution
        h.set_subharm(numpy.ones((3, 3, 3)), numpy.asarray(sh.fx, dtype=float), numpy.asarray(sh.fy, dtype=float),
                      numpy.asarray(sh.df, dtype=float))
        z = numpy.zeros((B, N, N))
        phs = h.screens_coeffs(z, z, numpy.ascontiguousarray(rand.real, dtype=float), numpy.ascontiguousarray(rand.imag, dtype=float))
    finally:
        h.close()
    return phs if double else phs[:B]


def generate_random_coefficients(shape):
    """fast/funcs.py:352-356: all real parts, then all imaginary parts, from the module generator that
    `Fast.set_seed` reseeds (shared with fast_amd.fast, like funcs._R in the reference)."""
    from . import fast as _fast
    return _fast._R.normal(0, 1, size=shape) + 1j * _fast._R.normal(0, 1, size=shape)


l_path = host.l_path
