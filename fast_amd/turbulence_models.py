"""Hufnagel-Valley 5/7 Cn2, Bufton wind and equivalent-layer compression.
Same functions and signatures as fast/turbulence_models.py:4-105 of the reference."""
import numpy


def HV57(h, w=21, A=1.7e-14):
    """Cn2 [m^-2/3] at heights h [m] (turbulence_models.py:4-19)."""
    h = numpy.asarray(h)
    upper = 0.00594 * (w / 27) ** 2 * (1e-5 * h) ** 10 * numpy.exp(-h / 1000)
    return upper + 2.7e-16 * numpy.exp(-h / 1500) + A * numpy.exp(-h / 100.)


def Bufton_wind(h, vg=8, vt=30, ht=9400., Lt=4800.):
    """Wind speed [m/s] (turbulence_models.py:22-38)."""
    h = numpy.asarray(h)
    return vg + vt * numpy.exp(-((h - ht) / Lt) ** 2)


def equivalent_layers(h, p, L, w=None):
    """Fusco (1999) equivalent layers: L equal-height slabs, cn2 summed, height (and wind)
    as the cn2-weighted 5/3 moment (turbulence_models.py:63-105)."""
    step = (h.max() - h.min()) / L
    slab = numpy.digitize(h, numpy.arange(h.min(), h.max(), step))
    h_out, c_out = numpy.zeros(L), numpy.zeros(L)
    w_out = numpy.zeros(L) if w is not None else None
    for i in range(L):
        sel = slab == i + 1
        tot = p[sel].sum()
        c_out[i] = tot
        h_out[i] = ((p[sel] * h[sel] ** (5 / 3)).sum() / tot) ** (3 / 5)
        if w is not None:
            w_out[i] = ((p[sel] * w[sel] ** (5 / 3)).sum() / tot) ** (3 / 5)
    if w is not None:
        return h_out, c_out, w_out
    return h_out, c_out


def HV57_Bufton_profile(N, w=21, A=1.7e-14, vg=8, vt=30, ht=9400., Lt=4800.):
    """N-layer profile from 1 m bins up to 30 km (turbulence_models.py:41-60)."""
    h0 = numpy.arange(0, 30000)
    return equivalent_layers(h0, HV57(h0, w, A), N, w=Bufton_wind(h0, vg, vt, ht, Lt))
