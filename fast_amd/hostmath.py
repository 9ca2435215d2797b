"""Host-side numerics the reference takes from `aotools` (absent in this image; restated
from that package's documented behaviour -- see SURVEY.md section 8c) and small helpers.

Call sites in the reference: aotools.circle funcs.py:263; gaussian2d funcs.py:290,340,348;
fouriertransform.ft2 funcs.py:309; cn2_to_r0 / isoplanaticAngle / coherenceTime /
rytov_variance fast.py:264-273; functions.zernike.zernIndex ao_power_spectra.py:11.
"""
import numpy as np
from scipy.integrate import simpson


def circle(radius, size):
    """Disc of `radius` pixels on a size x size grid, pixel centres at 0.5, 1.5, ...,
    disc centre at size/2 (aotools.circle, origin='middle')."""
    c = np.arange(0.5, size, 1.0) - size / 2.0
    cc = c * c                                   # x*x + y*y of the meshgrid form, without the two N x N copies
    return (cc[None, :] + cc[:, None] <= radius * radius).astype(float)


def gaussian2d(size, width):
    """exp(-((xc-X)^2 + (yc-Y)^2) / (2 width^2)), centre at size/2, image of shape `size`
    (rows, columns) (aotools.gaussian2d)."""
    try:
        ys, xs = size[0], size[1]
    except (TypeError, IndexError):
        xs = ys = size
    w = float(width)
    X, Y = np.meshgrid(np.arange(0, xs), np.arange(0, ys))
    return np.exp(-(((xs / 2.0 - X) / w) ** 2 + ((ys / 2.0 - Y) / w) ** 2) / 2)


def ft2(g, delta):
    """Centred 2-D DFT scaled by delta^2 (aotools.fouriertransform.ft2)."""
    return np.fft.fftshift(np.fft.fft2(np.fft.fftshift(g, axes=(-1, -2))), axes=(-1, -2)) * delta ** 2


def cn2_to_r0(cn2, lamda=500e-9):
    return (0.423 * (2 * np.pi / lamda) ** 2 * cn2) ** (-3.0 / 5.0)


def isoplanatic_angle(cn2, h, lamda=500e-9):
    """aotools.isoplanaticAngle: arcseconds (the reference stores it as theta0 / theta0_los and in the THETA0 card)."""
    return 0.057 * lamda ** (6.0 / 5.0) * np.sum(cn2 * h ** (5.0 / 3.0)) ** (-3.0 / 5.0) * 180.0 * 3600.0 / np.pi


def coherence_time(cn2, v, lamda=500e-9):
    return float(0.057 * lamda ** (6.0 / 5.0) * np.sum(cn2 * v ** (5.0 / 3.0)) ** (-3.0 / 5.0))


def rytov_variance(cn2, h, lamda=500e-9):
    k = 2 * np.pi / lamda
    return float(2.25 * k ** (7.0 / 6.0) * np.sum(cn2 * h ** (5.0 / 6.0)))


def noll_to_nm(j):
    n = int((-1.0 + np.sqrt(8 * (j - 1) + 1)) / 2.0)
    p = j - (n * (n + 1)) / 2.0
    k = n % 2
    m = int((p + k) / 2.0) * 2 - k
    if m != 0:
        m *= 1 if j % 2 == 0 else -1
    return n, m


_SIMPSON_CACHE = {}


def simpson_weights(f):
    """Weights w with simpson(y, x=f) == w @ y for every y (scipy's rule is linear in y).
    Used so that the GPU evaluates funcs.integrate_powerspectrum (funcs.py:100-115) as a
    weighted sum with exactly scipy's end-interval handling.  Cached per axis (sweeps)."""
    f = np.asarray(f, dtype=float)
    key = (len(f), float(f[0]), float(f[-1]))
    if key not in _SIMPSON_CACHE:
        if len(_SIMPSON_CACHE) > 16:
            _SIMPSON_CACHE.clear()
        _SIMPSON_CACHE[key] = simpson(np.eye(len(f)), x=f, axis=-1)
    return _SIMPSON_CACHE[key]
