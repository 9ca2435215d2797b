"""CPU oracle for the FAST Monte-Carlo hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A numpy restatement, as pure functions of explicit arrays, of the reference's
algorithm for SURVEY.md section 8 rows 1-12.  Each function cites the reference
file:line it follows (paths relative to /root/reference).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this package; `fast_amd/` never does.

Parity status: PINNED by outputs of the reference itself.  `tools/capture_golden/
capture.py` imports /root/reference/fast in the build container and stores its
inputs/outputs under `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks
every function here against them.  The reference's own tests hold no known-answer
vectors.  Fixtures whose values pass through the absent
third-party package `aotools` (pupil, fibre mode, pupil filter) were produced with
our stand-ins and are labelled "stand-in dependent" in tests/golden/MANIFEST.md;
the Monte-Carlo kernel fixtures (`kat_*`) are not.

Third-party arithmetic the reference calls and this oracle calls identically:
numpy.random.Generator.normal, numpy.fft, scipy.integrate.simpson, scipy.special.jv.
"""
import numpy as np
from scipy.integrate import simpson
from scipy.special import jv

TWO_PI = 2.0 * np.pi


# --------------------------------------------------------------------------
# Row 11: frequency grids                      fast/fast.py:830-844, 877-921
# --------------------------------------------------------------------------
class FreqGrid:
    """Angular spatial-frequency grid built from 1-D axes (fast/fast.py:877-921).

    `fx[i, j] = axis_x[j]`, `fy[i, j] = axis_y[i]` (numpy.meshgrid default 'xy').
    For stacked axes of shape (P, n) the grids are (P, n, n)  (fast.py:896-902).
    """

    def __init__(self, axis_x, axis_y=None):
        axis_x = np.asarray(axis_x, dtype=float)
        axis_y = axis_x if axis_y is None else np.asarray(axis_y, dtype=float)
        self.axis_x, self.axis_y = axis_x, axis_y
        self.df = axis_x[..., 1] - axis_x[..., 0]
        if axis_x.ndim == 1:
            self.fx, self.fy = np.meshgrid(axis_x, axis_y)
        else:
            self.fx = np.stack([np.meshgrid(ax, ay)[0] for ax, ay in zip(axis_x, axis_y)])
            self.fy = np.stack([np.meshgrid(ax, ay)[1] for ax, ay in zip(axis_x, axis_y)])
        self.fabs = np.sqrt(self.fx ** 2 + self.fy ** 2)


def main_grid(N, dx):
    """fast/fast.py:830-833: axis = arange(-N/2, N/2) * 2*pi/(N*dx)."""
    df = TWO_PI / (N * dx)
    return FreqGrid(np.arange(-N / 2.0, N / 2.0) * df)


def subharm_grid(N, dx, pmax=3):
    """fast/fast.py:835-844: per level p, axis = [-1, 0, 1] * 2*pi/(3^p * N*dx)."""
    D = dx * N
    axes = [np.arange(-1, 2) * (TWO_PI / (3 ** p * D)) for p in range(1, pmax + 1)]
    return FreqGrid(np.array(axes))


# --------------------------------------------------------------------------
# Row 6: von Karman spectrum                              fast/funcs.py:138-173
# --------------------------------------------------------------------------
def von_karman(fabs, cn2, L0, l0, C=TWO_PI):
    """0.033 cn2_l exp(-k^2/km^2) / (k^2 + k0^2)^(11/6), infinities -> 0 (funcs.py:152-170).

    Returns shape (L, *fabs.shape).
    """
    cn2 = np.atleast_1d(np.asarray(cn2, dtype=float))
    km = 5.92 / l0
    k0 = C / L0
    with np.errstate(divide="ignore", invalid="ignore"):
        base = 0.033 * np.exp(-fabs ** 2 / km ** 2) / (fabs ** 2 + k0 ** 2) ** (11 / 6.0)
    out = cn2.reshape((-1,) + (1,) * base.ndim) * base[np.newaxis]
    out[np.isinf(out)] = 0.0
    return out


# --------------------------------------------------------------------------
# Row 7: low-frequency (corrected-region) mask   fast/ao_power_spectra.py:10-21,54-76,119-141
# --------------------------------------------------------------------------
def noll_to_nm(j):
    """Noll index -> (n, m); aotools.functions.zernike.zernIndex (third party, absent)."""
    n = int((-1.0 + np.sqrt(8 * (j - 1) + 1)) / 2.0)
    p = j - (n * (n + 1)) / 2.0
    k = n % 2
    m = int((p + k) / 2.0) * 2 - k
    if m != 0:
        m *= 1 if j % 2 == 0 else -1
    return n, m


def zernike_sq_filter(fabs, fx, fy, D, n_noll):
    """sum_{j=1..n_noll} |Z~_j(kappa)|^2 with the centre pixel forced to 1
    (ao_power_spectra.py:54-76 with gamma=None, plusminus=False, n_noll_start=1)."""
    phi = np.arctan2(fy, fx)
    out = np.zeros(fabs.shape)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = fabs * D / 2
        for j in range(1, n_noll + 1):
            n, m = noll_to_nm(j)
            radial = 2 * jv(n + 1, x) / x
            if m == 0:
                z2 = (n + 1) * radial ** 2
            elif j % 2 == 0:
                z2 = 2 * (n + 1) * (radial * np.cos(m * phi)) ** 2
            else:
                z2 = 2 * (n + 1) * (radial * np.sin(m * phi)) ** 2
            out = out + z2
    out[..., int(fabs.shape[-2] / 2), int(fabs.shape[-1] / 2)] = 1
    return out


def mask_lf(fx, fy, d_wfs, modal=False, modal_mult=1, zmax=None, D=None):
    """ao_power_spectra.py:119-141 (Gtilt=False).  int64 0/1 for zonal / radial-modal,
    float64 in [0, 1] for the Zernike-modal case."""
    fmax = np.pi / d_wfs
    wfs_space = np.logical_and(np.abs(fx) <= fmax, np.abs(fy) <= fmax)
    if modal:
        fabs = np.sqrt(fx ** 2 + fy ** 2)
        if zmax is None:
            dm_space = fabs <= fmax * modal_mult
        else:
            dm_space = zernike_sq_filter(fabs, fx, fy, D, zmax)
    else:
        dm_space = wfs_space
    dm_space = np.where(dm_space < 1, dm_space, 1)
    return wfs_space * dm_space


# --------------------------------------------------------------------------
# Row 8: anisoplanatic / servo-lag transfer function   ao_power_spectra.py:225-270
# --------------------------------------------------------------------------
def g_ao(fx, fy, fabs, mask, mode, h, wind, dtheta, D_tx, zmax, t_loop, t_exp):
    """Per-layer (L, n, n) transfer function; the scalar 1 for 'NOAO' (235-236)."""
    if mode not in ("NOAO", "AO", "TT", "LGSAO"):
        raise Exception("Mode not recognised")
    if mode == "NOAO":
        return 1
    h = np.asarray(h, dtype=float)
    wind = np.asarray(wind, dtype=float)
    dr = np.outer(h, np.asarray(dtype=float, a=dtheta) / 206265.0)        # (L, 2)  line 245
    lead = (slice(None),) + (None,) * fx.ndim
    dr_k = fx[None] * dr[:, 0][lead] + fy[None] * dr[:, 1][lead]          # line 247
    v_k = fx[None] * wind[:, 0][lead] + fy[None] * wind[:, 1][lead]       # line 250
    s = np.sinc(t_exp * v_k / TWO_PI)                                     # line 255
    aniso = 1 - 2 * np.cos(dr_k - t_loop * v_k) * s + s ** 2              # lines 254-257
    if mode in ("AO", "TT"):
        return aniso * mask + (1 - mask)                                  # line 260
    s_l = np.sinc(t_exp * v_k / TWO_PI)                                   # LGSAO, 262-267
    aniso_lgs = 1 - 2 * np.cos(-t_loop * v_k) * s_l + s_l ** 2
    Z = zernike_sq_filter(fabs, fx, fy, D_tx, 4)
    return mask * (Z * aniso + (1 - Z) * aniso_lgs) + (1 - mask)


# --------------------------------------------------------------------------
# Row 9: WFS aliasing                               ao_power_spectra.py:163-223
# --------------------------------------------------------------------------
def alias_openloop(grid, d_wfs, cn2, mask, wind, t_exp, lmax, kmax, L0, l0):
    """Sum over the (2*lmax+1)(2*kmax+1)-1 shifted von Karman spectra, with the
    reference's special-cased centre row / column / pixel (208-213), times
    sinc^2(t_exp v.kappa / 2pi) * mask (216), NaN -> 0 (221).  Shape (L, n, n), or (L, 3, 3, 3) on the stacked
    sub-harmonic grids (fast.py:504-509)."""
    fx, fy, fabs = grid.fx, grid.fy, grid.fabs
    cn2 = np.asarray(cn2, dtype=float)
    wind = np.asarray(wind, dtype=float)
    mid_r = int(fx.shape[-2] / 2.0)
    mid_c = int(fy.shape[-1] / 2.0)
    lead = (slice(None),) + (None,) * fx.ndim              # (n, n) main grid or the stacked (3, 3, 3) sub-harmonic grids
    v_k = fx[None] * wind[:, 0][lead] + fy[None] * wind[:, 1][lead]
    alias = np.zeros((len(cn2),) + fabs.shape)
    with np.errstate(divide="ignore", invalid="ignore"):
        sinc2 = np.sinc(t_exp * v_k / TWO_PI) ** 2
        term_0 = fx ** 2 * fy ** 2 / fabs ** 4
        for l in range(-lmax, lmax + 1):
            for k in range(-kmax, kmax + 1):
                if l == 0 and k == 0:
                    continue
                sh = FreqGrid(grid.axis_x - TWO_PI * k / d_wfs, grid.axis_y - TWO_PI * l / d_wfs)
                term_1 = (fx / sh.fy + fy / sh.fx) ** 2
                term_2 = von_karman(sh.fabs, cn2, L0, l0)
                mult = term_1 * term_2 * term_0
                mult[..., mid_r, mid_c] = 0.0
                if l == 0:
                    mult[..., mid_r, :] = term_2[..., mid_r, :]
                if k == 0:
                    mult[..., mid_c] = term_2[..., mid_c]
                    mult[..., mid_r, mid_c] = term_2[..., mid_r, mid_c]
                alias += mult
        alias *= sinc2 * mask
    alias[np.isnan(alias)] = 0.0
    return alias


# --------------------------------------------------------------------------
# Row 10: noise, assembly, Simpson scalars, log-amplitude spectrum
# --------------------------------------------------------------------------
def noise_openloop(fx, fy, fabs, d_wfs, noise_var, mask):
    """ao_power_spectra.py:148-161 (freq_per_layer False)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        ps = noise_var / (fabs ** 2 * np.sinc(d_wfs * fx / TWO_PI) ** 2 * np.sinc(d_wfs * fy / TWO_PI) ** 2)
    ps[..., int(ps.shape[-2] / 2.0), int(ps.shape[-1] / 2.0)] = 0.0
    return mask * ps


def simpson2d(P, f):
    """funcs.py:100-115: scipy Simpson over the last axis, then over the next."""
    return simpson(simpson(P, x=f), x=f)


def logamp_spectrum(fabs, h, cn2, wvl, pupil_filter, L0, l0):
    """ao_power_spectra.py:272-301, ndarray pupil filter, layered path -> (n, n)."""
    h = np.asarray(h, dtype=float)
    ps = von_karman(fabs, cn2, L0, l0) * TWO_PI * (TWO_PI / wvl) ** 2
    ps = ps * np.sin(wvl * h[:, None, None] * fabs[None] ** 2 / (4 * np.pi)) ** 2
    if pupil_filter is not None:
        ps = ps * pupil_filter
    return ps.sum(0)


def residual_powerspec(N, dx, cn2, h, wind, L0, l0, wvl, ao_mode, d_wfs, dtheta, D_ground,
                       zmax, t_loop, t_exp, alias, noise, lf_mask, pupil_filter):
    """Fast.compute_powerspec, fast/fast.py:445-492 (main grid only).

    Returns a dict with `powerspec` (N, N), `powerspec_per_layer` (L, N, N) and the
    Simpson scalars.  `lf_mask` comes from mask_lf(); `pupil_filter` is (N, N)."""
    g = main_grid(N, dx)
    k = TWO_PI / wvl
    f = g.axis_x
    hf_mask = 1 - lf_mask
    turb = von_karman(g.fabs, cn2, L0, l0)                                      # 448-449
    G = g_ao(g.fx, g.fy, g.fabs, lf_mask, ao_mode, h, wind, dtheta, D_ground, zmax, t_loop, t_exp)
    out = {}
    out["aniso_servo_error"] = simpson2d((G * turb).sum(0) * lf_mask * TWO_PI * k ** 2, f)   # 456-457
    if alias and ao_mode != "NOAO":                                             # 459-468
        alias_ps = alias_openloop(g, d_wfs, cn2, lf_mask, wind, t_exp, 5, 5, L0, l0)
        out["alias_error"] = simpson2d((alias_ps * TWO_PI * k ** 2).sum(0), f)
    else:
        alias_ps = 0.0
        out["alias_error"] = 0.0
    if noise > 0 and ao_mode != "NOAO":                                         # 470-476
        noise_ps = noise_openloop(g.fx, g.fy, g.fabs, d_wfs, noise, lf_mask)
        out["noise_error"] = simpson2d(noise_ps, f)
    else:
        noise_ps = 0.0
        out["noise_error"] = 0.0
    per_layer = TWO_PI * k ** 2 * (turb * G + alias_ps) + noise_ps / len(h)      # 478-479
    ps = per_layer.sum(0)                                                       # 481
    out["powerspec_per_layer"] = per_layer
    out["powerspec"] = ps
    out["fitting_error"] = simpson2d(ps * hf_mask, f)                           # 483
    out["phs_var"] = simpson2d(ps, f)                                           # 484
    out["phs_var_weights"] = simpson2d(per_layer, f) / out["phs_var"]           # 485
    la = logamp_spectrum(g.fabs, h, cn2, wvl, pupil_filter, L0, l0)             # 490-491
    out["logamp_powerspec"] = la
    out["logamp_var"] = simpson2d(la, f)                                        # 492
    return out


def subharm_powerspec(N, dx, cn2, h, wind, L0, l0, wvl, ao_mode, d_wfs, dtheta, D_ground,
                      zmax, t_loop, t_exp, alias, noise, modal, modal_mult):
    """fast/fast.py:494-523: the same assembly on the (3, 3, 3) sub-harmonic grids."""
    g = subharm_grid(N, dx)
    k = TWO_PI / wvl
    mask = mask_lf(g.fx, g.fy, d_wfs, modal=modal, modal_mult=modal_mult, zmax=zmax, D=D_ground)
    turb = von_karman(g.fabs, cn2, L0, l0)                 # funcs.py:164 on (3, 3, 3) grids: the layer axis leads
    G = g_ao(g.fx, g.fy, g.fabs, mask, ao_mode, h, wind, dtheta, D_ground, zmax, t_loop, t_exp)
    if alias and ao_mode != "NOAO":
        alias_ps = alias_openloop(g, d_wfs, cn2, mask, wind, t_exp, 5, 5, L0, l0)
    else:
        alias_ps = 0.0
    if noise > 0 and ao_mode != "NOAO":
        noise_ps = noise_openloop(g.fx, g.fy, g.fabs, d_wfs, noise, mask)
    else:
        noise_ps = 0.0
    per_layer = TWO_PI * k ** 2 * (turb * G + alias_ps) + noise_ps / len(h)
    return per_layer.sum(0), g


# --------------------------------------------------------------------------
# Rows 1-5c: the Monte-Carlo realisation pipeline
# --------------------------------------------------------------------------
def draw_coefficients(rng, shape):
    """funcs.py:352-356: ALL real parts are drawn first, then all imaginary parts."""
    re = rng.normal(0, 1, size=shape)
    im = rng.normal(0, 1, size=shape)
    return re + 1j * im


def draw_logamp(rng, n_iter, logamp_var):
    """fast.py:639-645 -> funcs.py:358-365 (non-temporal): Re(N + iN) * sqrt(var).
    Consumes 2*n_iter normals; the imaginary draw is discarded but advances the stream."""
    re = rng.normal(0, 1, size=(n_iter,))
    rng.normal(0, 1, size=(n_iter,))
    return re * np.sqrt(logamp_var)


def screens_fftw(coloured, df):
    """funcs.py:212-215 + fast.py:431-438: fftshift, unnormalised FORWARD DFT over the
    last two axes, fftshift.  Canonical branch of this oracle."""
    x = np.fft.fftshift(coloured * df, axes=(-1, -2))
    return np.fft.fftshift(np.fft.fft2(x, axes=(-2, -1)), axes=(-1, -2))


def screens_numpy_branch(coloured, df):
    """funcs.py:218 -> aotools.fouriertransform.ift2(rand*df, 1) (third party, absent;
    restated from its published form: N = shape[0], ALL-axes ifftshift).  NOT pinned
    against the real aotools; kept for reference only."""
    N0 = coloured.shape[0]
    return np.fft.ifftshift(np.fft.ifft2(np.fft.ifftshift(coloured * df))) * (N0 * 1) ** 2


def double_screens(z):
    """funcs.py:220-221: vstack([Re, Im]) -- one complex FFT yields two real screens."""
    return np.vstack([z.real, z.imag])


def crop_lo(N, Np):
    """fast.py:390: first row/column of the pupil window."""
    return (N - Np) // 2


def crop(screens, N, Np):
    """fast.py:596 with pup_coords from fast.py:390."""
    lo, hi = (N - Np) // 2, (N + Np) // 2
    return screens[..., lo:hi, lo:hi]


def subharm_screens(rand_lo, sh_grid, N, dx):
    """funcs.py:225-258 (double=True): 3 levels x 3x3 modes evaluated on the full grid,
    per-screen complex mean removed, Re/Im stacked."""
    D = dx * N
    coords = np.arange(-D / 2, D / 2, dx)
    if len(coords) == N + 1:
        coords = coords[:-1]
    x, y = np.meshgrid(coords, coords)
    acc = np.zeros((rand_lo.shape[0], N, N), dtype=complex)
    for i in range(3):
        c = rand_lo[:, i] * sh_grid.df[i]                                   # (B,3,3)
        modes = np.exp(1j * (x[None, None] * sh_grid.fx[i][..., None, None]
                             + y[None, None] * sh_grid.fy[i][..., None, None]))   # (3,3,N,N)
        acc = acc + np.einsum("bij,ijxy->bxy", c, modes)
    acc = acc - acc.mean((1, 2))[:, None, None]
    return np.vstack([acc.real, acc.imag])


def detector(phs, W, dx, logamp_chunk, coherent=False):
    """fast.py:647-668: sum_pix W exp(i phi) dx^2 / (sum W dx^2) * exp(chi); |.|^2 unless coherent."""
    a = (W * np.exp(1j * phs)).sum((1, 2)) * dx ** 2
    a = np.exp(logamp_chunk) * a
    a = a / (W.sum() * dx ** 2)
    return a if coherent else np.abs(a) ** 2


def powers_from_coefficients(coeffs, powerspec, df, W, dx, logamp_chunk, coherent=False,
                             sub=None):
    """One chunk of Fast.run from explicit coefficients (fast.py:589-605, 647-668).

    coeffs: (B, N, N) complex standard normals -> 2B powers ordered [Re screens, Im screens].
    sub: optional (rand_lo (B,3,3,3) complex, powerspec_subharm (3,3,3), sh_grid)."""
    N = powerspec.shape[-1]
    Np = W.shape[-1]
    z = screens_fftw(coeffs * np.sqrt(powerspec), df)
    phs = crop(double_screens(z), N, Np)
    if sub is not None:
        rand_lo, ps_lo, sh_grid = sub
        phs = phs + crop(subharm_screens(rand_lo * np.sqrt(ps_lo), sh_grid, N, dx), N, Np)
    return detector(phs, W, dx, logamp_chunk, coherent)


def monte_carlo(seed_or_rng, n_iter, n_chunks, powerspec, df, W, dx, logamp_var,
                coherent=False, sub=None, return_coeffs=False):
    """Fast.run, fast.py:115-140 (non-temporal): log-amplitudes first, then per chunk
    draw -> colour -> FFT -> crop -> detector.  Draw order == the reference's, so the
    same numpy seed gives the same `_r`."""
    rng = seed_or_rng if isinstance(seed_or_rng, np.random.Generator) else np.random.default_rng(seed_or_rng)
    M = n_iter // n_chunks
    N = powerspec.shape[-1]
    chi = draw_logamp(rng, n_iter, logamp_var)
    out = np.zeros((n_chunks, M), dtype=complex if coherent else float)
    kept = []
    for c in range(n_chunks):
        coeffs = draw_coefficients(rng, (M // 2, N, N))
        sub_c = None
        if sub is not None:
            ps_lo, sh_grid = sub
            sub_c = (draw_coefficients(rng, (M // 2,) + ps_lo.shape), ps_lo, sh_grid)
        out[c] = powers_from_coefficients(coeffs, powerspec, df, W, dx, chi[c * M:(c + 1) * M],
                                          coherent, sub_c)
        if return_coeffs:
            kept.append(coeffs)
    if return_coeffs:
        return out.flatten(), chi, kept
    return out.flatten()


# --------------------------------------------------------------------------
# Row 12: result statistics                               fast/fast.py:949-983
# --------------------------------------------------------------------------
def result_stats(r, dl):
    p = dl * r
    return {
        "dB_rel": 10 * np.log10(r), "dB_abs": 10 * np.log10(r * dl), "dBm": 10 * np.log10(r * dl / 1e-3),
        "power": p, "scintillation_index": (r / r.mean()).var(), "avg_power_W": p.mean(),
    }


# --------------------------------------------------------------------------
# analytic mean irradiance                fast/fast.py:736-761
#   aotools.fouriertransform.ft2 / ift2 (third party, absent): centred DFTs scaled by delta^2
#   resp. (N delta_f)^2, restated from their documented behaviour (see capture_golden/shims).
# --------------------------------------------------------------------------
def ft2_centred(g, delta):
    return np.fft.fftshift(np.fft.fft2(np.fft.fftshift(g))) * delta ** 2


def ift2_centred(G, delta_f):
    return np.fft.ifftshift(np.fft.ifft2(np.fft.ifftshift(G))) * (G.shape[0] * delta_f) ** 2


def mean_irradiance(powerspec, W, dx, df, diffraction_limit, onaxis=True):
    N = powerspec.shape[0]
    pupil = np.zeros((N, N))
    pupil[:W.shape[0], :W.shape[1]] = W                                  # fast.py:739-741
    phs_otf = ift2_centred(powerspec, df)                                # 745
    phs_sf = phs_otf[N // 2, N // 2] - phs_otf                           # 746
    pupil_otf = ift2_centred(np.abs(ft2_centred(pupil, dx)) ** 2, df) / (2 * np.pi) ** 2   # 748-749
    otf = np.exp(-phs_sf) * pupil_otf                                    # 751
    psf = otf.sum().real * dx ** 2 if onaxis else ft2_centred(otf, dx).real               # 753-757
    return psf * (diffraction_limit / (pupil.sum() * dx ** 2) ** 2)      # 759-761


# --------------------------------------------------------------------------
# link metrics over the power vector      fast/comms.py:171-262
# --------------------------------------------------------------------------
def _fade_runs(I, threshold):
    """Run-length view of the fade mask: (start, length) of every maximal run of I < threshold."""
    m = np.asarray(I) < threshold
    edges = np.flatnonzero(np.diff(np.concatenate(([0], m.view(np.int8), [0]))))
    return edges[0::2], edges[1::2] - edges[0::2], m


def fade_prob(I, threshold, min_fades=30):                       # comms.py:171-177
    below = int((np.asarray(I) < threshold).sum())
    return below / len(I) if below >= min_fades else np.nan


def fade_dur(I, threshold, dt=1, min_fades=30):                  # comms.py:180-195
    """Mean length of the fades that both start and end inside the record: a run touching the
    first sample has no rising edge (diff == 1) and a run touching the last sample has not ended."""
    start, length, m = _fade_runs(I, threshold)
    keep = (start > 0) & (start + length < len(m))
    if keep.sum() < min_fades:
        return np.nan
    return length[keep].mean() * dt


def q_function(x):                                               # comms.py:258-262
    from scipy.special import erfc
    return 0.5 * erfc(np.asarray(x) / np.sqrt(2.0))


def ber_ook(EbN0, samples=None):                                 # comms.py:198-222
    snr = np.sqrt(10 ** (EbN0 / 10))
    s = 1.0 if samples is None else samples / np.mean(samples)
    return np.mean(q_function(s * snr))


def sep_qam(M, EsN0, samples=None):                              # comms.py:225-242
    s = 1.0 if samples is None else samples / np.mean(samples)
    pf = (np.sqrt(M) - 1) / np.sqrt(M)
    q = q_function(np.sqrt(3 / (M - 1) * 10 ** (EsN0 / 10) * s ** 2))
    return 4 * np.mean(pf * q - pf ** 2 * q ** 2)


def ber_qam(M, EbN0, samples=None):                              # comms.py:245-255
    return sep_qam(M, 10 * np.log10(np.log2(M)) + EbN0, samples) / np.log2(M)


# --------------------------------------------------------------------------
# Temporal (frozen-flow) mode -- SURVEY 8f rank 2
#   fast/fast.py:846-864 (frequencies), 394-405 (high-resolution pupil filter),
#   538-587 (shifts, temporal log-amplitude spectrum), 607-637 (shifted screens),
#   fast/funcs.py:367-375 (coloured log-amplitude series)
# --------------------------------------------------------------------------
from scipy.interpolate import RectBivariateSpline  # noqa: E402  (third party the reference calls)


def temporal_freqs(n_layers, Ny, Nx, wind_speed, wind_dir_deg, dt, dfy):
    """fast.py:846-864: per layer, x axis in LINEAR frequency 1/(Nx v dt), y axis = main dfy grid,
    rotated by the wind direction.  Returns axes_x (L,Nx), axes_y (L,Ny), fabs (L,Ny,Nx)."""
    ax, ay, fabs = [], [], []
    for i in range(n_layers):
        dft = 1 / (Nx * wind_speed[i] * dt)
        fx_axis = np.arange(-Nx / 2, Nx / 2) * dft
        fy_axis = np.arange(-Ny / 2, Ny / 2) * dfy
        fx, fy = np.meshgrid(fx_axis, fy_axis)
        rot = np.radians(wind_dir_deg[i])
        fxr = fx * np.cos(rot) - fy * np.sin(rot)
        fyr = fx * np.sin(rot) + fy * np.cos(rot)
        ax.append(fx_axis)
        ay.append(fy_axis)
        fabs.append(np.sqrt(fxr ** 2 + fyr ** 2))
    return np.array(ax), np.array(ay), np.array(fabs)


def _aperture(N, dx, D, obsc, Ny=None):
    """funcs.compute_pupil (funcs.py:261-277) with aotools.circle restated (third party)."""
    def circle(radius, size):
        c = np.arange(0.5, size, 1.0) - size / 2.0
        x, y = np.meshgrid(c, c)
        return (x * x + y * y <= radius * radius).astype(float)
    ap = circle(D / dx / 2, N) - circle(obsc / dx / 2, N)
    if Ny is not None:
        if Ny > N:
            pad = (Ny - N) // 2
            ap = np.pad(ap, [(0, 0), (pad, pad)])
        if Ny < N:
            cut = (N - Ny) // 2
            ap = ap[:, cut:-cut]
    return ap / np.sqrt(ap.sum() * dx ** 2)


def temporal_pupil_filter(ax_t, ay_t, df_main, D, obsc, W0, Np, dx):
    """fast.py:394-405: pupil filter on a grid fine enough for the temporal frequencies, as a
    bilinear RectBivariateSpline (funcs.py:308-315, spline=True)."""
    f_max = max(ax_t.max(), ay_t.max())
    dx_req = np.pi / f_max
    N_req = int(2 * np.ceil(2 * np.pi / (df_main * dx_req) / 2))
    Ny = 2 * Np
    pupil = _aperture(N_req, dx_req, D, obsc, Ny=Ny)
    ys, xs = pupil.shape
    Xg, Yg = np.meshgrid(np.arange(xs), np.arange(ys))
    w = W0 / dx_req / np.sqrt(2)
    mode = np.exp(-(((xs / 2.0 - Xg) / w) ** 2 + ((ys / 2.0 - Yg) / w) ** 2) / 2) * np.sqrt(2 / (np.pi * W0 ** 2)) / pupil.max()
    field = pupil * mode
    P = np.abs(np.fft.fftshift(np.fft.fft2(np.fft.fftshift(field, axes=(-1, -2))), axes=(-1, -2))) ** 2
    P = P / field.sum() ** 2
    fx_axis = np.arange(-N_req / 2., N_req / 2.) * (TWO_PI / (N_req * dx_req))
    fy_axis = np.arange(-Ny / 2., Ny / 2.) * (TWO_PI / (Ny * dx))
    return RectBivariateSpline(fx_axis, fy_axis, P, kx=1, ky=1, s=0)


def temporal_logamp_spectrum(ax_t, ay_t, fabs_t, h, cn2, wvl, spline, L0, l0, dfy):
    """fast.py:582-587 with ao_power_spectra.py:272-301 (freq_per_layer, spline pupil filter)."""
    h = np.asarray(h, dtype=float)
    cn2 = np.asarray(cn2, dtype=float)
    km, k0 = 5.92 / l0, TWO_PI / L0
    with np.errstate(divide="ignore", invalid="ignore"):
        vk = (0.033 * np.exp(-fabs_t ** 2 / km ** 2) / (fabs_t ** 2 + k0 ** 2) ** (11 / 6.)) * cn2[:, None, None]
    vk[np.isinf(vk)] = 0.
    ps = vk * TWO_PI * (TWO_PI / wvl) ** 2
    ps = ps * np.sin(wvl * h[:, None, None] * fabs_t ** 2 / (4 * np.pi)) ** 2
    P = np.stack([spline(ay_t[i], ax_t[i]) for i in range(len(h))])
    return (ps * P).sum(0).sum(-2) * dfy


def draw_logamp_temporal(rng, n_iter, logamp_var, tps):
    """funcs.py:367-375: coloured series = centred DFT of white noise * sqrt(normalised spectrum)."""
    r = rng.normal(0, 1, size=(n_iter,)) + 1j * rng.normal(0, 1, size=(n_iter,))
    r = r * np.sqrt(tps / tps.sum())
    series = np.fft.fftshift(np.fft.fft(np.fft.fftshift(r)))
    return (series.T * np.sqrt(logamp_var)).real


def temporal_pixel_shifts(M, dt, wind_vector, dx):
    """fast.py:543-544 -> (L, 2, M)."""
    dts = np.arange(1, M + 1) * dt
    return dts * np.asarray(wind_vector)[..., np.newaxis] / dx


def temporal_coords(interp_coords, N):
    """fast.py:621-626: wrapped + sorted sample coordinates and the roll that undoes the sort."""
    coord = interp_coords % N
    coord = np.sort(coord, axis=-1)
    diffs = np.abs(np.diff(coord, axis=-1))
    shifts = diffs.argmax(-1)
    shifts[np.isclose(diffs, 1).all(-1)] = 0
    return coord, shifts


def temporal_chunk_phases(scrns, coord, shifts, Np):
    """fast.py:628-633: bilinear samples of each layer's screen at the shifted pupil grid, summed."""
    L, _, M, _ = coord.shape
    N = scrns.shape[-1]
    interps = [RectBivariateSpline(np.arange(N), np.arange(N), s, kx=1, ky=1, s=0) for s in scrns]
    phs = np.zeros((M, Np, Np))
    for i in range(L):
        for j in range(M):
            p = interps[i](coord[i, 0, j], coord[i, 1, j])
            phs[j] += np.roll(p, -shifts[i, :, j], axis=(0, 1))
    return phs


def monte_carlo_temporal(seed_or_rng, n_iter, n_chunks, per_layer, df, W, dx, logamp_var, tps, wind_vector, dt, N, Np,
                         coherent=False, return_screens=False):
    """Fast.run with TEMPORAL (fast.py:115-140, 607-637), FFTW-branch screens (double=False)."""
    rng = seed_or_rng if isinstance(seed_or_rng, np.random.Generator) else np.random.default_rng(seed_or_rng)
    M = n_iter // n_chunks
    chi = draw_logamp_temporal(rng, n_iter, logamp_var, tps)
    coeffs = draw_coefficients(rng, per_layer.shape)
    scrns = screens_fftw(coeffs * np.sqrt(per_layer), df).real
    lo = (N - Np) // 2
    pup = np.arange(lo, lo + Np).astype(float)
    shifts_px = temporal_pixel_shifts(M, dt, wind_vector, dx)
    interp = np.stack([pup, pup])[None, :, None, :] + shifts_px[:, :, :, None]
    out = np.zeros((n_chunks, M), dtype=complex if coherent else float)
    for c in range(n_chunks):
        coord, shifts = temporal_coords(interp, N)
        phs = temporal_chunk_phases(scrns, coord, shifts, Np)
        out[c] = detector(phs, W, dx, chi[c * M:(c + 1) * M], coherent)
        interp = interp + shifts_px[:, :, -1, None, None]
    if return_screens:
        return out.flatten(), chi, scrns
    return out.flatten()
