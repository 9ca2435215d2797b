"""The opt-in host path of `fast_amd.Fast` (`GPU_FALLBACK: True`): numpy in place of libfastmc.so when the library or a GPU is missing.

The reference degrades the same way when its accelerator is missing: `fast/fast.py:106-110` warns "pyfftw not found, falling back
to numpy" and carries on.  By DEFAULT `fast_amd` does not -- it raises, so that nothing measured or tested can have come from a
fall-back -- but BASELINE configs[0] is literally "CPU numpy FFT path (plumbing, no GPU)", and a sweep driver on a login node should
be able to build its objects.  With `GPU_FALLBACK: True` the object logs the reference's kind of warning and every library call goes
to this module instead: the same `Handle` interface (`HostHandle`), the same power-spectrum entry points, float64 numpy throughout,
draws by numpy in the reference's order (there is no device generator without a device: `GPU_RNG` is 'host' here whatever was asked).

This is product code and imports nothing from `oracle/` (tests/test_lib_abi.py enforces it); it is an independent statement of the
reference's formulas, each citing file:line, and tests/test_hostpath.py holds it to the reference's own fixtures on the CPU.
bench.py and the `-m gpu` tests never enable it."""
import logging

import numpy as np
from scipy.interpolate import RectBivariateSpline
from scipy.special import jv

from . import host
from . import hostmath as hm

logger = logging.getLogger(__name__)
TWO_PI = 2 * np.pi


# ------------------------------------------------------------------ power spectrum (fast.py:445-492)
def _spectrum(N, dx, wvl, L0, l0, ao_mode, alias, noise, d_wfs, t_loop, t_exp, dtheta, cn2, h, wind, pupil_filter, simpson_w,
              lf_mask=None, modal=False, modal_mult=1, zmax=None, D_ground=0.0):
    """Residual phase PSD on the main grid, per layer, with its terms: ao_power_spectra.py:119-301, funcs.py:138-173,
    fast.py:445-492.  (fast_amd/host.py: subharm_spectrum is the same arithmetic on the 27 sub-harmonic frequencies.)"""
    cn2, h, wind = np.asarray(cn2, float), np.asarray(h, float), np.asarray(wind, float)
    L = len(cn2)
    axis = host.freq_axis(N, dx)
    fx, fy, fabs = host.mesh(axis)
    c = N // 2
    k = TWO_PI / wvl
    mask = host.lf_mask(fx, fy, d_wfs, modal, modal_mult, zmax, D_ground) if lf_mask is None else np.asarray(lf_mask)
    mask_out = mask.astype(float)
    turb = host._von_karman(fabs, cn2, L0, l0)                                   # funcs.py:138-173
    v_k = fx[None] * wind[:, 0][:, None, None] + fy[None] * wind[:, 1][:, None, None]
    if ao_mode == 'NOAO':                                                        # ao_power_spectra.py:235-236
        G = np.ones((L, N, N))
    else:
        dr = np.outer(h, np.asarray(dtheta, float) / 206265.)                    # :245
        dr_k = fx[None] * dr[:, 0][:, None, None] + fy[None] * dr[:, 1][:, None, None]
        s = np.sinc(t_exp * v_k / TWO_PI)
        aniso = 1 - 2 * np.cos(dr_k - t_loop * v_k) * s + s ** 2                 # :249-256
        if ao_mode in ('AO', 'TT'):
            G = aniso * mask + (1 - mask)
        else:                                                                     # LGSAO :262-267
            aniso_l = 1 - 2 * np.cos(-t_loop * v_k) * s + s ** 2
            Z = host.zernike_sq(fabs, fx, fy, D_ground, 4)
            G = mask * (Z * aniso + (1 - Z) * aniso_l) + (1 - mask)
    al = np.zeros((L, N, N))
    if alias and ao_mode != 'NOAO':                                              # ao_power_spectra.py:163-223, lmax = kmax = 5
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = fx ** 2 * fy ** 2 / fabs ** 4
            for l in range(-5, 6):
                for kk in range(-5, 6):
                    if l == 0 and kk == 0:
                        continue
                    sx, sy = np.meshgrid(axis - TWO_PI * kk / d_wfs, axis - TWO_PI * l / d_wfs)
                    t2 = host._von_karman(np.sqrt(sx ** 2 + sy ** 2), cn2, L0, l0)
                    m = (fx / sy + fy / sx) ** 2 * t2 * t0
                    m[..., c, c] = 0.
                    if l == 0:
                        m[..., c, :] = t2[..., c, :]
                    if kk == 0:
                        m[..., c] = t2[..., c]
                        m[..., c, c] = t2[..., c, c]
                    al += m
            al *= np.sinc(t_exp * v_k / TWO_PI) ** 2 * mask
        al[np.isnan(al)] = 0.
    no = np.zeros((N, N))
    if noise > 0 and ao_mode != 'NOAO':                                          # ao_power_spectra.py:148-161
        with np.errstate(divide="ignore", invalid="ignore"):
            no = noise / (fabs ** 2 * np.sinc(d_wfs * fx / TWO_PI) ** 2 * np.sinc(d_wfs * fy / TWO_PI) ** 2)
        no[c, c] = 0.
        no = mask * no
    per_layer = TWO_PI * k ** 2 * (turb * G + al) + no / L                       # fast.py:478-479
    ps = per_layer.sum(0)

    def integ(g):                                                                # funcs.py:100-115: 2-D Simpson over the axis
        return float(simpson_w @ g @ simpson_w)
    la = None
    logamp_var = 0.0
    if pupil_filter is not None:                                                 # ao_power_spectra.py:272-301
        pf = np.asarray(pupil_filter, float)
        la = (turb * (TWO_PI * k ** 2) * np.sin(wvl * h[:, None, None] * fabs[None] ** 2 / (4 * np.pi)) ** 2).sum(0) * pf
        logamp_var = integ(la)
    out = {"powerspec": ps, "powerspec_per_layer": per_layer, "logamp_powerspec": la if la is not None else np.zeros((N, N)),
           "lf_mask": mask_out, "turb_powerspec": turb, "G_ao": G, "alias_powerspec": al, "noise_powerspec": no}
    hf = 1 - mask_out
    pl_var = np.array([integ(p) for p in per_layer])
    phs_var = integ(ps)
    out.update({"phs_var": phs_var, "fitting_error": integ(ps * hf),                                       # fast.py:483-485
                "aniso_servo_error": integ((G * turb).sum(0) * mask_out * TWO_PI * k ** 2),                # :456-457
                "alias_error": integ((al * TWO_PI * k ** 2).sum(0)), "noise_error": integ(no),            # :464-465, 473
                "logamp_var": logamp_var, "phs_var_weights": pl_var / phs_var, "kernel_ms": 0.0})
    return out


def powerspec(N, *args, per_layer=False, device=None, lgs_z=None, **kw):
    out = _spectrum(N, *args, **kw)
    if not per_layer:
        out["powerspec_per_layer"] = None
    return out


def powerspec_terms(N, *args, device=None, lgs_z=None, **kw):
    out = _spectrum(N, *args, **kw)
    return {k: out[k] for k in ("turb_powerspec", "G_ao", "alias_powerspec", "noise_powerspec")}


def powerspec_set(handle, df, *args, pupil_filter_token=0, lgs_z=None, **kw):
    out = _spectrum(handle.N, *args, **kw)
    handle._grids = {k: out[k] for k in ("powerspec", "logamp_powerspec", "lf_mask")}
    handle.set_spectrum(out["powerspec"], df)
    return {k: out[k] for k in ("aniso_servo_error", "alias_error", "noise_error", "fitting_error", "phs_var", "logamp_var",
                                "phs_var_weights", "kernel_ms")}


def centred_fft2(g, device=None, inverse=False):
    """_lib.centred_fft2 in numpy: fftshift(fft2(fftshift(g))), or ifftshift(ifft2(ifftshift(g))) (host.mean_irradiance)."""
    g = np.asarray(g)
    if inverse:
        return np.fft.ifftshift(np.fft.ifft2(np.fft.ifftshift(g)))
    return np.fft.fftshift(np.fft.fft2(np.fft.fftshift(g)))


# ------------------------------------------------------------------ the handle (rows 1-5c, TEMPORAL, post-processing)
class HostHandle:
    """`fast_amd._lib.Handle` in numpy: what `Fast` and `multi.DeviceGroup` call, nothing else."""
    precision = "f64"
    precision_requested = "f64"

    def __init__(self, N, Np, precision="f64", device=None):
        self.N, self.Np, self.device = int(N), int(Np), -1
        self._amp = self._W = self._sh = self._results = self._layers = None
        self._grids = {}
        self._lo, self._dx, self._df = 0, 1.0, 1.0

    def close(self):
        pass

    # problem
    def set_spectrum(self, powerspec, df):
        ps = np.asarray(powerspec, float)
        if not np.isfinite(ps).all() or (ps < 0).any():
            raise ValueError("powerspec must be finite and non-negative")
        self._amp, self._df = np.sqrt(ps), float(df)

    def set_pupil(self, W, crop_lo, dx):
        self._W, self._lo, self._dx = np.asarray(W, float), int(crop_lo), float(dx)

    def set_subharm(self, powerspec_sh, fx=None, fy=None, df=None):
        self._sh = None if powerspec_sh is None else (np.sqrt(np.asarray(powerspec_sh, float)), np.asarray(fx, float), np.asarray(fy, float),
                                                      np.asarray(df, float))

    def set_batch(self, batch):
        pass

    def set_rng_precision(self, precision):
        pass

    def kernel_path(self, force=-1):
        return 1                       # (no kernels: the FFT is numpy's; Fast only uses this for a performance hint)

    def last_kernels(self):
        return "numpy.fft (host path)", "numpy (host path)"

    def last_clock(self):
        return None

    def effective_precision(self):
        return "f64"

    def last_timing(self):
        return {"total_ms": 0.0, "rows_ms": 0.0, "cols_ms": 0.0, "finalize_ms": 0.0, "rows_launches": 0, "cols_launches": 0, "finalize_launches": 0}

    def powerspec_get(self, which):
        return self._grids[which]

    # rows 3, 4, 5c: screens of a chunk (funcs.py:210-258, fast.py:589-605)
    def screens_coeffs(self, coeff_re, coeff_im, sh_re=None, sh_im=None):
        N, Np, lo = self.N, self.Np, self._lo
        rand = (np.asarray(coeff_re) + 1j * np.asarray(coeff_im)) * self._amp
        z = np.fft.fftshift(np.fft.fft2(np.fft.fftshift(rand * self._df, axes=(-1, -2))), axes=(-1, -2))      # funcs.py:212-215
        phs = np.vstack([z.real, z.imag])[:, lo:lo + Np, lo:lo + Np]                                            # :220-221, fast.py:596
        if self._sh is not None and sh_re is not None:
            amp, fx, fy, df = self._sh
            rl = (np.asarray(sh_re) + 1j * np.asarray(sh_im)) * amp
            key = (lo, Np, self._dx)
            if getattr(self, "_sh_key", None) != key:
                # funcs.py:225-258: 27 modes exp(i (x fx + y fy)) on the FULL grid, the screen's mean removed; only the window
                # is kept (fast.py:603) -- mode minus its own full-grid mean, by linearity, cached per window
                D = self._dx * N
                coords = np.arange(-D / 2, D / 2, self._dx)[:N]
                xx, yy = np.meshgrid(coords, coords)
                modes = np.empty((3, 3, 3, Np, Np), dtype=complex)
                for p in range(3):
                    for i in range(3):
                        for j in range(3):
                            full = np.exp(1j * (xx * fx[p, i, j] + yy * fy[p, i, j]))
                            modes[p, i, j] = full[lo:lo + Np, lo:lo + Np] - full.mean()
                self._sh_modes, self._sh_key = modes, key
            lo_scr = np.einsum("bpij,pijyx->byx", rl * df[None, :, None, None], self._sh_modes)
            phs = phs + np.vstack([lo_scr.real, lo_scr.imag])
        return phs

    def _detect(self, phs, logamp, coherent):                     # fast.py:647-668
        W = self._W
        a = (W * np.exp(1j * phs)).sum((1, 2)) * self._dx ** 2 * np.exp(np.asarray(logamp)) / (W.sum() * self._dx ** 2)
        self._results = a if coherent else np.abs(a) ** 2
        return self._results

    def run_coeffs(self, coeff_re, coeff_im, logamp, coherent=False, sh_re=None, sh_im=None):
        return self._detect(self.screens_coeffs(coeff_re, coeff_im, sh_re, sh_im), logamp, coherent)

    def run(self, *a, **k):
        raise RuntimeError("the host path has no device generator: draws are numpy's (GPU_RNG 'host')")

    run_async = screens = rng_coeffs = rng_logamp = run

    # TEMPORAL (fast.py:607-637)
    def set_layer_screens(self, screens):
        ax = np.arange(self.N)
        self._layers = [RectBivariateSpline(ax, ax, s, kx=1, ky=1, s=0) for s in np.asarray(screens, float)]

    def temporal_phases(self, xs, ys, roll):
        L, M, Np = np.asarray(xs).shape
        phs = np.zeros((M, Np, Np))
        for i, scrn in enumerate(self._layers):
            for j in range(M):
                phs[j] += np.roll(scrn(xs[i, j], ys[i, j]), -np.asarray(roll)[i, :, j], axis=(0, 1))
        return phs

    def temporal_chunk(self, xs, ys, roll, logamp, coherent=False):
        return self._detect(self.temporal_phases(xs, ys, roll), logamp, coherent)

    # post-processing on the resident results (fast.py:949-983)
    def set_results(self, values):
        self._results = np.asarray(values)

    def _powers(self):
        if self._results is None:
            raise RuntimeError("no run results")
        r = self._results
        return np.abs(r) ** 2 if np.iscomplexobj(r) else r

    def histogram(self, lo_db, hi_db, nbins):
        db = 10 * np.log10(self._powers())
        out = np.zeros(nbins + 2, dtype=np.int64)
        idx = np.floor((db - lo_db) / (hi_db - lo_db) * nbins).astype(np.int64)
        out[0] = (db < lo_db).sum()
        out[-1] = (db >= hi_db).sum()
        inside = (db >= lo_db) & (db < hi_db)
        np.add.at(out, 1 + np.clip(idx[inside], 0, nbins - 1), 1)
        return out

    def result_stats(self, thresholds=()):
        r = self._powers()
        mean = r.mean()
        return {"n": r.size, "mean": mean, "scintillation_index": (r ** 2).mean() / mean ** 2 - 1.0, "mean_dB_rel": (10 * np.log10(r)).mean(),
                "avg_dB_rel": 10 * np.log10(mean), "min": r.min(), "max": r.max(),
                "fade_prob": np.array([(r < t).mean() for t in np.asarray(thresholds, float).ravel()])}


Handle = HostHandle          # (the name fast_amd.fast asks its backend for)


def unavailable_reason():
    """None when libfastmc.so loads and sees a GPU, else why not (the text of the warning)."""
    from . import _lib
    try:
        n = _lib.device_count()
    except Exception as e:              # library not built / not loadable
        return f"libfastmc.so is not available ({e})"
    return None if n > 0 else "no gfx950 GPU is visible"
