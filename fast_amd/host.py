"""Host-side, one-off producers of the GPU kernels' inputs: atmosphere scalars, grid sizing,
pupil and fibre mode, low-frequency mask, link budget, sub-harmonic spectrum.

These are the "host-side producers" of SURVEY.md section 8a: cheap numpy work done once per
`Fast` object, restated from the reference's init sequence (fast/fast.py:95-104), each
function citing the lines it follows.  Nothing here is on the Monte-Carlo hot path and
nothing here needs a GPU, so it is unit-tested on CPU against the golden fixtures.
"""
import logging
from types import SimpleNamespace

import numpy as np
from scipy.optimize import minimize_scalar
from scipy.special import jv

from . import hostmath as hm

logger = logging.getLogger(__name__)
TWO_PI = 2 * np.pi
# grid sizes served by the wave-FFT kernels: 64 * 2^k * {1, 3, 5, 7, 9}, 128 ... 2048 (fmc_wavefft.h), and 4096
# (4 interleaved sub-rows of 1024; 2048 runs as 2)
WAVE_FFT_SIZES = sorted(64 * q * 2 ** k for q in (1, 3, 5, 7, 9) for k in range(6) if 2 <= q * 2 ** k <= 32) + [4096]
MAX_NPXLS = 4096          # libfastmc: every N <= 4096 has a kernel family (fastmc_create) ...
MAX_NPXLS_SUBROWS = 8192  # ... and the grids N = 64 P S / 50 P S of S <= 8 sub-rows (P <= 24 values per lane) go on to 8192


def _radix_ok(q):
    if q < 2 or q > 32:
        return False
    while q % 2 == 0:
        q //= 2
    return q in (1, 3, 5, 7, 9)


def big_grid_supported(N):
    """Grids beyond 4096 that libfastmc serves (fmc_core.h: wave_rt_split / mr_split with up to eight sub-rows): N = 64 P S or
    50 P S, S <= 8, 7 <= P <= 24, P = 2^k times 1, 3, 5, 7 or 9 -- 4608, 4800, 5000, 5120, 6000, 6144, 6400, 7000, 7168, 7680,
    8000, 8192 ..."""
    if N <= MAX_NPXLS or N > MAX_NPXLS_SUBROWS:
        return False
    for unit in (64, 50):
        if N % unit == 0:
            q = N // unit
            if any(q % S == 0 and 7 <= q // S <= 24 and _radix_ok(q // S) for S in range(2, 9)):
                return True
    return False
# Measured throughput [k iterations/s, Np = 82, float64 pipeline, float64 device generator] of every grid size that has a pruned-FFT row
# (tools/sizesweep.sh on one MI355X: profiles/r06y_sizesweep_every_multiple_of_64.txt, round 6): every multiple of
# 64 (packed rows 128 / 256 / 512, packed sub-rows 192 ... 4096, the dense 1024 row) and the 50 P S grids of the 50-lane kernels.
# GPU_ROUND_NPXLS (off by default) rounds an auto-sized grid up to the smallest of these that is within 10 % of the fastest one not
# smaller than it (the 50-lane grids run 0.55-0.69 of the 1024-point row's rate per pixel and lose to the next multiple of 64).
FAST_SIZE_RATE = {
    128: 13104, 192: 7811, 256: 5870, 320: 3892, 384: 3015, 448: 2414, 512: 2104, 576: 1624, 640: 1341, 704: 1118,
    768: 1033, 832: 856, 896: 738, 960: 664, 1024: 617, 1088: 506, 1152: 458, 1216: 420, 1280: 407, 1344: 348, 1408: 318,
    1472: 293, 1536: 288, 1600: 242, 1664: 223, 1728: 200, 1792: 214, 1856: 169, 1920: 166, 1984: 158, 2048: 168,
    2112: 125, 2176: 131, 2240: 119, 2304: 130, 2368: 113, 2432: 107, 2496: 95.1, 2560: 105, 2624: 88.9, 2688: 86.8,
    2752: 81.8, 2816: 85.6, 2880: 68.5, 2944: 71.4, 3008: 65.3, 3072: 73.2, 3136: 62.1, 3200: 61.6, 3264: 60.2, 3328: 62.5,
    3392: 52.9, 3456: 52.4, 3520: 51.0, 3584: 54.1, 3648: 42.6, 3712: 46.4, 3776: 41.3, 3840: 47.3, 3904: 40.3, 3968: 39.9,
    4032: 39.1, 4096: 41.8,
    100: 8395, 150: 6179, 200: 4958, 250: 3865, 300: 3337, 350: 2581, 400: 2264, 450: 1856, 500: 1695, 600: 1239, 700: 726,
    800: 617, 900: 546, 1000: 415, 1200: 277, 1400: 184, 1500: 199, 2000: 96.7, 2500: 70.2, 3000: 43.7, 4000: 24.8,
}
ROUND_UP_SIZES = sorted(FAST_SIZE_RATE)
_ROUND_WARNED = False


def round_up_size(N):
    """The grid size GPU_ROUND_NPXLS gives an auto-sized N: None when nothing fast is >= N."""
    cand = [s for s in ROUND_UP_SIZES if s >= N]
    if not cand:
        return None
    best = max(FAST_SIZE_RATE[s] for s in cand)
    return next(s for s in cand if FAST_SIZE_RATE[s] >= 0.9 * best)


# ----------------------------------------------------------------------------- geometry
def l_path(h_sat, zeta):
    """Slant range to a satellite at altitude h_sat, zenith angle zeta [deg] (funcs.py:388-399)."""
    r_e = 6.371009e6
    z = np.radians(zeta)
    b = -2 * r_e * np.cos(np.pi - z)
    c = r_e ** 2 - (r_e + h_sat) ** 2
    disc = np.sqrt(b ** 2 - 4 * c)
    r1 = (-b + disc) / 2
    return r1 if r1 >= 0 else (-b - disc) / 2


def wind_correction(h, theta_loop, t_loop):
    """Apparent wind from the satellite's angular motion (funcs.py:403-406)."""
    return -np.array([np.sin(np.radians(theta_loop[0] / 3600)) * h / t_loop,
                      np.sin(np.radians(theta_loop[1] / 3600)) * h / t_loop]).T


def atmosphere(p):
    """Fast.init_atmos (fast.py:229-276)."""
    a = SimpleNamespace()
    a.zenith_correction = 1 / np.cos(np.radians(p['ZENITH_ANGLE']))            # fast.py:763-766
    a.h = p['H_TURB'] * a.zenith_correction
    a.cn2 = p['CN2_TURB'] * a.zenith_correction
    a.L = p['L_SAT'] if p['L_SAT'] is not None else l_path(p['H_SAT'], p['ZENITH_ANGLE'])
    a.dtheta = p['DTHETA']
    a.paa = np.sqrt(a.dtheta[0] ** 2 + a.dtheta[1] ** 2)
    wind_dir = p['WIND_DIR']
    if 'AZIMUT_SAT' in p:
        wind_dir = [(x - p['AZIMUT_SAT']) % 380 for x in wind_dir]              # sic: % 380 (fast.py:250)
    a.wind_dir = wind_dir
    ang = np.radians(wind_dir)
    a.wind_vector = (p['WIND_SPD'] * np.array([np.cos(ang), np.sin(ang) / a.zenith_correction])).T
    if 'ANISO_DL' in p:
        a.wind_correction = wind_correction(a.h, p['ANISO_DL'], p['TLOOP'])
        a.wind_vector = a.wind_vector + a.wind_correction
    a.wind_speed = np.sqrt(a.wind_vector[:, 0] ** 2 + a.wind_vector[:, 1] ** 2)
    a.r0 = hm.cn2_to_r0(p['CN2_TURB'].sum(), lamda=500e-9)
    a.theta0 = hm.isoplanatic_angle(p['CN2_TURB'], p['H_TURB'], lamda=500e-9)
    a.tau0 = hm.coherence_time(p['CN2_TURB'], p['WIND_SPD'], lamda=500e-9)
    a.rytov_variance = hm.rytov_variance(p['CN2_TURB'], p['H_TURB'], lamda=500e-9)
    a.r0_los = hm.cn2_to_r0(a.cn2.sum(), lamda=p['WVL'])
    a.theta0_los = hm.isoplanatic_angle(a.cn2, a.h, lamda=p['WVL'])
    a.tau0_los = hm.coherence_time(a.cn2, a.wind_speed, lamda=p['WVL'])
    a.rytov_variance_los = hm.rytov_variance(a.cn2, a.h, lamda=p['WVL'])
    return a


def grid_size(p, atm, size_limit=True):
    """Pixel scale, grid size and pupil-window size (Fast.init_frequency_grid, fast.py:147-211)."""
    D = p['D_GROUND']
    if p['DX'] == 'auto':
        dx = np.min([p['DSUBAP'] / 2, atm.r0_los / 2, D / 10])
        if p['AO_MODE'] == 'NOAO':
            dx = atm.r0_los / 2
        logger.info(f"Auto set DX to {dx}")
    else:
        dx = p['DX']
    if p['NPXLS'] == 'auto':
        nyq = np.min([np.pi / (atm.h[-1] * atm.paa / 206265.),
                      np.pi / (max(atm.wind_speed) * p['TLOOP']),
                      np.pi / p['DSUBAP'] / 5])
        n_nyq = int(2 * np.ceil(2 * np.pi / (nyq * dx) / 2))
        n_ap = int(2 * np.ceil(D / dx / 2)) + 2
        n_t = int(p['WIND_SPD'].max() * p['DT'] * p['NITER'] / p['DX'] / 2) if p['TEMPORAL'] else 0
        N = np.max([n_nyq, n_ap, n_t])
        rnd = p.get('GPU_ROUND_NPXLS', False)
        if rnd == 'auto':
            # host-drawn coefficients (parity with the reference for a seed) and TEMPORAL series (numpy
            # draws in the reference's order) need the reference's exact grid; the device generator does not
            rnd = p.get('GPU_RNG', 'device') == 'device' and not p['TEMPORAL']
        if rnd:
            # the reference's auto rule is a lower bound and gives arbitrary even sizes (164 for the shipped
            # example; 256 on the packed-row kernels is faster than anything in between); the next size of the
            # fast kernel family samples the spectrum slightly finer
            bigger = round_up_size(N)
            if bigger and bigger != N:
                # a drop-in user comparing headers with the reference's would trip on a silent change: said once per process as a
                # warning (the reference's own auto rule would give N), afterwards at INFO
                global _ROUND_WARNED
                (logger.info if _ROUND_WARNED else logger.warning)(
                    f"GPU_ROUND_NPXLS: auto NPXLS {N} (the reference's rule) -> {bigger} (the next fast kernel size; "
                    f"GPU_ROUND_NPXLS: False keeps {N})")
                _ROUND_WARNED = True
                N = bigger
        logger.info(f"Auto set NPXLS to {N}")
        if p['AO_MODE'] == 'NOAO' and not np.isinf(p['L0']):
            n_l0 = int(2 * np.ceil((p['L0'] * 2) / dx) / 2)
            if n_l0 > N:
                logger.warning(f"L0 set with NOAO mode, low orders may be undersampled. Recommended NPXLS: {n_l0}")
    else:
        N = p['NPXLS']
    if N > 2048:
        logger.warning(f"NPXLS is large ({N}) and may cause very high memory usage")
    if size_limit and (N > MAX_NPXLS_SUBROWS or (N > MAX_NPXLS and not big_grid_supported(N) and int(np.ceil(D / dx)) + 2 > 256)):
        # fail before any O(N^2) host work; a TEMPORAL series auto-sizes to half its total wind
        # displacement (fast.py:201-206), which outgrows the kernels quickly
        hint = " (TEMPORAL: fewer steps per object, a shorter DT or a coarser DX)" if p['TEMPORAL'] else ""
        raise Exception(f"NPXLS = {N} exceeds the GPU kernels' limit of {MAX_NPXLS_SUBROWS if N > MAX_NPXLS_SUBROWS else MAX_NPXLS} "
                        f"(every N <= {MAX_NPXLS}; up to {MAX_NPXLS_SUBROWS} with a pupil window of at most 256 pixels, or as a grid of "
                        f"up to eight sub-rows, N = 64 P S or 50 P S: 4608, 5000, 5120, 6000, 6144, 7000, 8000, 8192 ...){hint}")
    Np = int(np.ceil(D / dx)) + 2
    return dx, int(N), Np


def freq_axis(N, dx):
    """arange(-N/2, N/2) * 2 pi / (N dx)  (fast.py:830-833)."""
    return np.arange(-N / 2., N / 2.) * (TWO_PI / (N * dx))


def subharm_axes(N, dx, pmax=3):
    """(3, 3) axes [-1, 0, 1] * 2 pi / (3^p N dx)  (fast.py:835-844)."""
    D = dx * N
    return np.array([np.arange(-1, 2) * (TWO_PI / (3 ** p * D)) for p in range(1, pmax + 1)])


def mesh(axis):
    """fx, fy, |f| for a 1-D axis or a stack of axes (fast.py:896-921)."""
    if axis.ndim == 1:
        fx, fy = np.meshgrid(axis, axis)
    else:
        fx = np.stack([np.meshgrid(a, a)[0] for a in axis])
        fy = np.stack([np.meshgrid(a, a)[1] for a in axis])
    return fx, fy, np.sqrt(fx ** 2 + fy ** 2)


# ----------------------------------------------------------------------------- masks
def zernike_sq(fabs, fx, fy, D, n_noll):
    """sum_j |Z~_j|^2, centre forced to 1 (ao_power_spectra.py:10-21, 54-76)."""
    phi = np.arctan2(fy, fx)
    out = np.zeros(fabs.shape)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = fabs * D / 2
        for j in range(1, n_noll + 1):
            n, m = hm.noll_to_nm(j)
            rad = 2 * jv(n + 1, x) / x
            if m == 0:
                out = out + (n + 1) * rad ** 2
            elif j % 2 == 0:
                out = out + 2 * (n + 1) * (rad * np.cos(m * phi)) ** 2
            else:
                out = out + 2 * (n + 1) * (rad * np.sin(m * phi)) ** 2
    out[..., int(fabs.shape[-2] / 2), int(fabs.shape[-1] / 2)] = 1
    return out


def lf_mask(fx, fy, d_wfs, modal, modal_mult, zmax, D):
    """Corrected region of the AO system (ao_power_spectra.mask_lf, 119-141).  Host version: used for
    the 27 sub-harmonic frequencies only; the N x N mask is evaluated by the GPU kernel."""
    fmax = np.pi / d_wfs
    wfs = np.logical_and(np.abs(fx) <= fmax, np.abs(fy) <= fmax)
    if not modal:
        dm = wfs
    elif zmax is None:
        dm = np.sqrt(fx ** 2 + fy ** 2) <= fmax * modal_mult
    else:
        dm = zernike_sq(np.sqrt(fx ** 2 + fy ** 2), fx, fy, D, zmax)
    return wfs * np.where(dm < 1, dm, 1)


# ----------------------------------------------------------------------------- pupil / fibre mode
def aperture(N, dx, D, obsc=0):
    """Unit-power annular aperture (funcs.compute_pupil, funcs.py:261-277, Ny=None)."""
    ap = hm.circle(D / dx / 2, N) - hm.circle(obsc / dx / 2, N)
    return ap / np.sqrt(ap.sum() * dx ** 2)


def _support_box(a):
    """Bounding box (r0, r1, c0, c1) of the non-zero pixels of a 2-D array."""
    rows, cols = np.flatnonzero(a.any(axis=1)), np.flatnonzero(a.any(axis=0))
    return rows[0], rows[-1] + 1, cols[0], cols[-1] + 1


def _gaussian_box(shape, width, box):
    """hm.gaussian2d(shape, width)[r0:r1, c0:c1], the same arithmetic per pixel, without the rest of the image."""
    r0, r1, c0, c1 = box
    w = float(width)
    X, Y = np.meshgrid(np.arange(c0, c1), np.arange(r0, r1))
    return np.exp(-(((shape[1] / 2.0 - X) / w) ** 2 + ((shape[0] / 2.0 - Y) / w) ** 2) / 2)


def coupling_loss(W, shape, pupil, dx, box=None):
    """funcs.py:347-350.  The overlap integral only sees the pupil's support: with `box` (its bounding
    box) the Gaussian is evaluated there alone -- the Brent search calls this ~40 times and the reference
    forms an N x N Gaussian each time (2 s at N = 2048).  The products are summed in a full-size array
    of zeros, so that the value (numpy's pairwise summation order included) is bit-identical to the
    reference's and the search takes the same path."""
    if box is None:
        field = hm.gaussian2d(shape, W / dx / np.sqrt(2)) * np.sqrt(2. / (np.pi * W ** 2))
        return 1 - np.abs((field * pupil).sum() * dx ** 2) ** 2
    r0, r1, c0, c1 = box
    prod = np.zeros(shape)
    prod[r0:r1, c0:c1] = _gaussian_box(shape, W / dx / np.sqrt(2), box) * np.sqrt(2. / (np.pi * W ** 2)) * pupil[r0:r1, c0:c1]
    return 1 - np.abs(prod.sum() * dx ** 2) ** 2


def best_gaussian(pupil, dx):
    """Brent search for the fibre-mode radius (funcs.optimize_fibre, funcs.py:317-345)."""
    shape = pupil.shape
    lo, hi = dx, max(shape) * dx
    box = _support_box(pupil)
    f = lambda W: coupling_loss(W, shape, pupil, dx, box)
    opt = minimize_scalar(f, bracket=[lo, hi]).x
    if abs(opt) < dx:
        logger.info("Gaussian mode optimisation failed, trying with different parameters")
        opt = minimize_scalar(f, bracket=[lo, 2 * hi]).x
        if abs(opt) < dx:
            raise Exception("Cannot optimise gaussian mode, try changing DX?")
    g = hm.gaussian2d(shape, opt / dx / np.sqrt(2)) * np.sqrt(2. / (np.pi * opt ** 2))
    return g, np.abs(opt)


def fibre_mode(pupil, dx, W0, D=None, obsc=None, ptype='gauss'):
    """funcs.compute_gaussian_mode (funcs.py:280-305) -> (mode, W0)."""
    nx, ny = pupil.shape
    if ptype == 'gauss':
        if W0 == "opt":
            g, opt = best_gaussian(pupil, dx)
            return g / pupil.max(), opt
        return hm.gaussian2d((nx, ny), W0 / dx / np.sqrt(2)) * np.sqrt(2 / (np.pi * W0 ** 2)) / pupil.max(), W0
    if ptype == 'axicon':
        if W0 == "opt":
            raise TypeError("Using 'axicon' and W0='opt' not supported, please set a value for W0")
        x = np.arange(-nx / 2, nx / 2, 1) * dx
        y = np.arange(-ny / 2, ny / 2, 1) * dx
        xx, yy = np.meshgrid(y, x)
        r = np.sqrt(xx ** 2 + yy ** 2)
        ring = np.exp(-(r - (obsc / 2 + (D / 2 - obsc / 2) / 2)) ** 2 / W0 ** 2)
        return ring / np.sqrt((ring ** 2).sum() * dx ** 2) / pupil.max(), W0
    raise Exception('ptype must be one of "gauss" or "axicon"')


def pupil_filter(field):
    """|FT(field)|^2 / (sum field)^2 (funcs.pupil_filter, funcs.py:308-315, spline=False)."""
    P = np.abs(hm.ft2(field, 1)) ** 2
    return P / field.sum() ** 2


_PUPIL_CACHE = {}
_PUPIL_TOKEN = 0


def pupils(p, N, Np, dx):
    """Fast.init_pupil_mask (fast.py:332-392), non-temporal part.  The result depends only on
    the aperture / grid parameters, so sweeps over atmospheric geometry reuse it."""
    key = (N, Np, float(dx), p['D_GROUND'], p['OBSC_GROUND'], p['D_SAT'], p['OBSC_SAT'], str(p['W0']), bool(p['AXICON']))
    if key in _PUPIL_CACHE:
        return _PUPIL_CACHE[key]
    # built locally and stored only when complete: an exception part-way (axicon with W0 'opt', a failed Brent
    # search) must not leave a half-filled entry for the next Fast() with the same key
    o = SimpleNamespace()
    D, obsc = p['D_GROUND'], p['OBSC_GROUND']
    o.dx_sat = p['D_SAT'] / 32
    full = aperture(N, dx, D, obsc)
    o.pupil_sat = aperture(32, o.dx_sat, p['D_SAT'], p['OBSC_SAT'])
    mode_full, o.W0 = fibre_mode(full, dx, p['W0'], D=D, obsc=obsc, ptype='axicon' if p['AXICON'] else 'gauss')
    o.pupil_mode_sat, o.W0_sat = fibre_mode(o.pupil_sat, o.dx_sat, "opt", ptype="gauss")
    o.pupil_filter = pupil_filter(full * mode_full)
    lo, hi = (N - Np) // 2, (N + Np) // 2
    o.crop_lo = lo
    o.pup_coords = np.array((np.arange(lo, hi), np.arange(lo, hi))).astype(int)
    o.pupil = full[lo:hi, lo:hi]
    o.pupil_mode = mode_full[lo:hi, lo:hi]
    # the arrays are shared by every Fast object of this geometry: read-only, so that an in-place edit of
    # `sim.pupil` cannot leak into later simulations (copy it to modify it)
    for a in (o.pupil, o.pupil_mode, o.pupil_sat, o.pupil_mode_sat, o.pupil_filter, o.pup_coords):
        a.flags.writeable = False
    global _PUPIL_TOKEN
    _PUPIL_TOKEN += 1
    o.token = _PUPIL_TOKEN          # names pupil_filter for the device-side cache (fastmc_ps_params.pupil_filter_token)
    if len(_PUPIL_CACHE) > 8:
        _PUPIL_CACHE.clear()
    _PUPIL_CACHE[key] = o
    return o


# ----------------------------------------------------------------------------- analytic mean irradiance
def mean_irradiance(powerspec, W, dx, df, diffraction_limit, onaxis=True, device=None, backend=None):
    """Fast.compute_mean_irradiance (fast.py:736-761): mean coupled flux from the optical transfer
    functions, no Monte Carlo.  Its three or four N x N transforms (aotools ft2 / ift2: centred DFTs
    scaled by delta^2 resp. (N delta_f)^2) run on the GPU (_lib.centred_fft2)."""
    if backend is None:
        from . import _lib as backend
    N = powerspec.shape[0]
    ft2 = lambda g, delta: backend.centred_fft2(g, device) * delta ** 2
    ift2 = lambda G, delta_f: backend.centred_fft2(G, device, inverse=True) * (N * delta_f) ** 2
    pupil = np.zeros(powerspec.shape)
    pupil[:W.shape[0], :W.shape[1]] = W
    phs_otf = ift2(powerspec, df)
    phs_sf = phs_otf[N // 2, N // 2] - phs_otf
    pupil_ft = ft2(pupil, dx)
    pupil_otf = ift2(np.abs(pupil_ft) ** 2, df) / (2 * np.pi) ** 2
    otf = np.exp(-phs_sf) * pupil_otf
    psf = otf.sum().real * dx ** 2 if onaxis else ft2(otf, dx).real
    return psf * (diffraction_limit / (pupil.sum() * dx ** 2) ** 2)


# ----------------------------------------------------------------------------- temporal (frozen-flow) mode
def aperture_rect(N, dx, D, obsc, Ny):
    """compute_pupil with Ny != N: columns zero-padded or cropped (funcs.py:265-273)."""
    ap = hm.circle(D / dx / 2, N) - hm.circle(obsc / dx / 2, N)
    if Ny > N:
        pad = (Ny - N) // 2
        ap = np.pad(ap, [(0, 0), (pad, pad)])
    if Ny < N:
        cut = (N - Ny) // 2
        ap = ap[:, cut:-cut]
    return ap / np.sqrt(ap.sum() * dx ** 2)


def temporal_setup(prob):
    """Everything TEMPORAL adds to the init (fast.py:217-219, 394-405, 538-587): per-layer temporal
    frequency grids, the high-resolution pupil-filter spline, the temporal log-amplitude spectrum
    and the per-step pixel shifts.  O(L * N * NITER) numpy work, once per object."""
    from scipy.interpolate import RectBivariateSpline
    p, atm = prob.params, prob.atm
    t = SimpleNamespace()
    L, Ny, Nx, dt = len(atm.h), prob.N, prob.Niter, p['DT']
    ax, ay, fabs = [], [], []
    for i in range(L):                                                    # fast.py:851-864
        fx_axis = np.arange(-Nx / 2, Nx / 2) * (1 / (Nx * atm.wind_speed[i] * dt))   # linear frequency (sic)
        fy_axis = np.arange(-Ny / 2, Ny / 2) * prob.df
        fx, fy = np.meshgrid(fx_axis, fy_axis)
        rot = np.radians(atm.wind_dir[i])
        fabs.append(np.sqrt((fx * np.cos(rot) - fy * np.sin(rot)) ** 2 + (fx * np.sin(rot) + fy * np.cos(rot)) ** 2))
        ax.append(fx_axis)
        ay.append(fy_axis)
    ax, ay, fabs = np.array(ax), np.array(ay), np.array(fabs)
    # pupil filter fine enough for those frequencies, as a bilinear spline (fast.py:396-405)
    dx_req = np.pi / max(ax.max(), ay.max())
    N_req = int(2 * np.ceil(2 * np.pi / (prob.df * dx_req) / 2))
    Nyp = 2 * prob.Np
    pupil_t = aperture_rect(N_req, dx_req, p['D_GROUND'], p['OBSC_GROUND'], Nyp)
    mode_t, _ = fibre_mode(pupil_t, dx_req, W0=prob.pup.W0, ptype="gauss")
    P = pupil_filter(pupil_t * mode_t)
    fxa = np.arange(-N_req / 2., N_req / 2.) * (TWO_PI / (N_req * dx_req))
    fya = np.arange(-Nyp / 2., Nyp / 2.) * (TWO_PI / (Nyp * prob.dx))
    spline = RectBivariateSpline(fxa, fya, P, kx=1, ky=1, s=0)
    # temporal log-amplitude spectrum (ao_power_spectra.py:272-301 on the per-layer grids; fast.py:582-587)
    turb = np.stack([_von_karman(fabs[i], [atm.cn2[i]], p['L0'], p['l0'])[0] for i in range(L)])
    ps = turb * TWO_PI * (TWO_PI / prob.wvl) ** 2
    ps = ps * np.sin(prob.wvl * atm.h[:, None, None] * fabs ** 2 / (4 * np.pi)) ** 2
    ps = ps * np.stack([spline(ay[i], ax[i]) for i in range(L)])
    t.logamp_powerspec = ps.sum(0).sum(-2) * prob.df
    t.pixel_shifts = (np.arange(1, prob.M + 1) * dt) * atm.wind_vector[..., np.newaxis] / prob.dx   # fast.py:543-544
    return t


def temporal_coords(interp_coords, N):
    """Wrapped + sorted sample coordinates and the roll that undoes the sort (fast.py:621-626)."""
    coord = np.sort(interp_coords % N, axis=-1)
    diffs = np.abs(np.diff(coord, axis=-1))
    shifts = diffs.argmax(-1)
    shifts[np.isclose(diffs, 1).all(-1)] = 0
    return coord, shifts


# ----------------------------------------------------------------------------- link budget
def link_budget(p, pup, atm, dx):
    """Fast.compute_link_budget (fast.py:670-734) -> (dict, diffraction_limit [W])."""
    wvl = p['WVL']
    if p['PROP_DIR'] == "up":
        D_t, D_r, obsc_t, obsc_r = p['D_GROUND'], p['D_SAT'], p['OBSC_GROUND'], p['OBSC_SAT']
        mode, dx_r, pupil_r, w0 = pup.pupil_mode_sat, pup.dx_sat, pup.pupil_sat, pup.W0
    else:
        D_t, D_r, obsc_t, obsc_r = p['D_SAT'], p['D_GROUND'], p['OBSC_SAT'], p['OBSC_GROUND']
        mode, dx_r, pupil_r, w0 = pup.pupil_mode, dx, pup.pupil, pup.W0_sat
    lb = {}
    lb['power'] = 10 * np.log10(p['POWER'] / 1e-3)
    lb['free_space'] = 10 * np.log10((wvl / (4 * np.pi * atm.L)) ** 2)
    alpha, gamma = D_t / (2 * w0), obsc_t / D_t
    g_t = 2 / alpha ** 2 * (np.exp(-alpha ** 2) - np.exp(-gamma ** 2 * alpha ** 2)) ** 2
    lb['transmitter_gain'] = 10 * np.log10((np.pi * D_t ** 2) * 4 * np.pi / wvl ** 2 * g_t)
    area = np.pi * ((D_r / 2) ** 2 - (obsc_r / 2) ** 2)
    lb['receiver_gain'] = 10 * np.log10(4 * np.pi * area / wvl ** 2)
    lb['transmission_loss'] = 10 * np.log10(p['TRANSMISSION'])
    lb['smf_coupling'] = 10 * np.log10(((pupil_r * mode).sum() * dx_r) ** 2 / (mode ** 2).sum())
    return lb, 10 ** (sum(lb.values()) / 10) / 1e3


# ----------------------------------------------------------------------------- sub-harmonic spectrum
def _von_karman(fabs, cn2, L0, l0):
    km, k0 = 5.92 / l0, TWO_PI / L0
    with np.errstate(divide="ignore", invalid="ignore"):
        base = 0.033 * np.exp(-fabs ** 2 / km ** 2) / (fabs ** 2 + k0 ** 2) ** (11 / 6.)
    out = np.asarray(cn2, dtype=float).reshape((-1,) + (1,) * base.ndim) * base[None]
    out[np.isinf(out)] = 0.
    return out


def subharm_spectrum(prob):
    """Residual PSD on the 27 sub-harmonic frequencies (fast.py:494-523): 3 x 3 x 3 points,
    evaluated on the host (the N x N main grid is evaluated on the GPU)."""
    p, atm = prob.params, prob.atm
    axes = subharm_axes(prob.N, prob.dx)
    fx, fy, fabs = mesh(axes)
    k = TWO_PI / p['WVL']
    L = len(atm.h)
    mask = lf_mask(fx, fy, prob.d_wfs, prob.modal, prob.modal_mult, prob.zmax, p['D_GROUND'])
    turb = _von_karman(fabs, atm.cn2, p['L0'], p['l0'])
    lead = (slice(None),) + (None,) * fx.ndim
    v_k = fx[None] * atm.wind_vector[:, 0][lead] + fy[None] * atm.wind_vector[:, 1][lead]
    mode = prob.ao_mode
    if mode == 'NOAO':
        G = 1
    else:
        dr = np.outer(atm.h, np.asarray(atm.dtheta, dtype=float) / 206265.)
        dr_k = fx[None] * dr[:, 0][lead] + fy[None] * dr[:, 1][lead]
        s = np.sinc(p['TEXP'] * v_k / TWO_PI)
        aniso = 1 - 2 * np.cos(dr_k - p['TLOOP'] * v_k) * s + s ** 2
        if mode in ('AO', 'TT'):
            G = aniso * mask + (1 - mask)
        else:
            aniso_l = 1 - 2 * np.cos(-p['TLOOP'] * v_k) * s + s ** 2
            Z = zernike_sq(fabs, fx, fy, p['D_GROUND'], 4)
            G = mask * (Z * aniso + (1 - Z) * aniso_l) + (1 - mask)
    alias = 0.
    if p['ALIAS'] and mode != 'NOAO':
        alias = np.zeros((L,) + fabs.shape)
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = fx ** 2 * fy ** 2 / fabs ** 4
            for l in range(-5, 6):
                for kk in range(-5, 6):
                    if l == 0 and kk == 0:
                        continue
                    sx, sy, sabs = _shifted(axes, kk, l, prob.d_wfs)
                    t2 = _von_karman(sabs, atm.cn2, p['L0'], p['l0'])
                    m = (fx / sy + fy / sx) ** 2 * t2 * t0
                    m[..., 1, 1] = 0.
                    if l == 0:
                        m[..., 1, :] = t2[..., 1, :]
                    if kk == 0:
                        m[..., 1] = t2[..., 1]
                        m[..., 1, 1] = t2[..., 1, 1]
                    alias += m
            alias *= np.sinc(p['TEXP'] * v_k / TWO_PI) ** 2 * mask
        alias[np.isnan(alias)] = 0.
    noise = 0.
    if p['NOISE'] > 0 and mode != 'NOAO':
        with np.errstate(divide="ignore", invalid="ignore"):
            noise = p['NOISE'] / (fabs ** 2 * np.sinc(prob.d_wfs * fx / TWO_PI) ** 2 * np.sinc(prob.d_wfs * fy / TWO_PI) ** 2)
        noise[..., 1, 1] = 0.
        noise = mask * noise
    per_layer = TWO_PI * k ** 2 * (turb * G + alias) + noise / L
    df = axes[..., 1] - axes[..., 0]
    # the bookkeeping the reference leaves on the object (fast.py:494-526)
    phs_var = per_layer.sum((-1, -2)) * df ** 2
    extra = SimpleNamespace(per_layer=per_layer, lf_mask=mask, turb=turb, G=G, alias=alias, noise=noise,
                            phs_var=phs_var, phs_var_weights=phs_var / phs_var.sum())
    return per_layer.sum(0), fx, fy, df, extra


def _shifted(axes, k, l, d):
    ax = axes - TWO_PI * k / d
    ay = axes - TWO_PI * l / d
    sx = np.stack([np.meshgrid(a, b)[0] for a, b in zip(ax, ay)])
    sy = np.stack([np.meshgrid(a, b)[1] for a, b in zip(ax, ay)])
    return sx, sy, np.sqrt(sx ** 2 + sy ** 2)


# ----------------------------------------------------------------------------- the whole host init
def build_problem(params, size_limit=True):
    """Everything `Fast.__init__` computes on the host before the GPU is needed
    (fast.py:71-103 minus compute_powerspec)."""
    p = params
    prob = SimpleNamespace(params=p)
    prob.Niter, prob.Nchunks = p['NITER'], p['NCHUNKS']
    if prob.Niter % prob.Nchunks != 0:
        raise Exception('NCHUNKS must divide NITER without remainder')
    prob.M = prob.Niter // prob.Nchunks
    if prob.M % 2 != 0 and not p['TEMPORAL']:
        raise Exception('NITER/NCHUNKS must be even number')
    prob.atm = atmosphere(p)
    prob.wvl = p['WVL']
    prob.k = TWO_PI / prob.wvl
    prob.dx, prob.N, prob.Np = grid_size(p, prob.atm, size_limit)
    prob.axis = freq_axis(prob.N, prob.dx)
    prob.df = prob.axis[1] - prob.axis[0]
    prob.subharm = bool(p['SUBHARM']) and not p['TEMPORAL']
    # AO parameters (fast.py:297-315)
    prob.ao_mode, prob.d_wfs = p['AO_MODE'], p['DSUBAP']
    prob.zmax, prob.modal, prob.modal_mult = p['ZMAX'], p['MODAL'], p['MODAL_MULT']
    if prob.ao_mode == 'TT':
        prob.zmax, prob.modal, prob.modal_mult = 3, True, 1
    if prob.ao_mode not in ('NOAO', 'AO', 'TT', 'LGSAO'):
        raise Exception('Mode not recognised, note that "AO_PA", "TT_PA" and "LGS_PA" are now "AO" and "TT" and "LGSAO')
    prob.pup = pupils(p, prob.N, prob.Np, prob.dx)
    prob.W = prob.pup.pupil * prob.pup.pupil_mode
    prob.link_budget, prob.diffraction_limit = link_budget(p, prob.pup, prob.atm, prob.dx)
    prob.simpson_w = hm.simpson_weights(prob.axis)
    prob.temporal = temporal_setup(prob) if p['TEMPORAL'] else None
    return prob
