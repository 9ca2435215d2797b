"""Helpers shared by the GPU parity tests (tests/test_gpu_parity_*.py; one file until round 5).

GPU parity: the HIP path (through the C-ABI) against the oracle and the golden fixtures.

Tolerances (identical coefficients): f64 pipeline rtol 1e-9 on per-iteration power and 1e-11 of
the screen's peak on phase; f32 pipeline rtol 1e-4 (atol 1e-9) on power.  Device-generator mode:
DEVICE_RTOL = 1e-5 on power, against the oracle fed with the device's own float32 draws coloured
in float32 as the kernels colour them (measured: ~1e-7), and against the oracle fed with the float64
restatement of the generator (oracle/devrng.py; the float32 hardware log2 / sqrt / sin / cos and the
float32 colouring then show: measured ~1e-7 too on few-radian screens)."""
import numpy as np
import pytest

from conftest import E2E_CASES, load_golden, params_from_json
import fast_amd
from fast_amd import _lib
from oracle import fastref as R
from oracle import devrng

__all__ = ["np", "pytest", "E2E_CASES", "load_golden", "params_from_json", "fast_amd", "_lib", "R", "devrng", "DEVICE_RTOL",
           "f32_draw_handle", "_oracle_powers_from_device_draws", "_oracle_powers_from_restated_draws", "_vk_spectrum", "_window_W", "_ps_call", "_small_problem"]

def f32_draw_handle(N, Np, precision="f64", device=0):
    """A handle with the OPT-IN float32 draw selected (fastmc_set_rng_precision FASTMC_F32) -- the generator the device-mode
    parity tests of rounds 1-4 were written against and still pin: 49 row variants x 2 precisions against the oracle on the
    restated float32 draws.  Since round 5 `fastmc_create` leaves a float64 handle drawing at the reference's precision
    (tests: test_handle_default_generator_is_the_reference_precision, the float64-generator tests, smoke()); the tests that
    concern that generator select it explicitly, so either starting point serves them."""
    h = _lib.Handle(N, Np, precision, device)
    h.set_rng_precision("f32")
    return h


DEVICE_RTOL = 1e-5       # device-generator powers vs the oracle (see the module docstring)


def _oracle_powers_from_device_draws(h, seed, real0, n, ps, df, W, lo, dx, logamp_var):
    """What fastmc_run must return for realisations [real0, real0 + n): the device's OWN coefficients
    (fastmc_rng_coeffs: float32 Box-Muller values) coloured in float32 with float32(sqrt(powerspec) df) as
    fmc_kernels.h:draw_coloured does, then the ORACLE's transform (funcs.py:212-215), crop at `lo` (fast.py:596),
    detector with the device's own log-amplitude normals (fast.py:647-668)."""
    N, Np = ps.shape[0], W.shape[0]
    amp32 = (np.sqrt(ps) * df).astype(np.float32)
    re, im = [], []
    for j in range(n):
        c = h.rng_coeffs(seed, real0 + j)
        cr = (c.real.astype(np.float32) * amp32).astype(np.float64)
        ci = (c.imag.astype(np.float32) * amp32).astype(np.float64)
        z = R.screens_fftw(cr + 1j * ci, 1.0)[lo:lo + Np, lo:lo + Np]
        re.append(z.real)
        im.append(z.imag)
    phs = np.stack(re + im)
    chi = h.rng_logamp(seed, 2 * real0, 2 * n) * np.sqrt(logamp_var)
    la = np.concatenate([chi[0::2], chi[1::2]])
    return R.detector(phs, W, dx, la)


def _oracle_powers_from_restated_draws(seed, real0, n, ps, df, W, lo, dx, logamp_var):
    """The same, with nothing read back from the device: the draws are the ORACLE's float64 restatement of the generator
    (oracle/devrng.device_coefficients, pinned to Random123 / xoshiro known answers), coloured in float64.  The device draws
    with hardware float32 log / sqrt / sin / cos and colours in float32: the two agree to ~1e-7 per coefficient."""
    N, Np = ps.shape[0], W.shape[0]
    amp = np.sqrt(ps) * df
    re, im = [], []
    for j in range(n):
        z = R.screens_fftw(devrng.device_coefficients(seed, real0 + j, N) * amp, 1.0)[lo:lo + Np, lo:lo + Np]
        re.append(z.real)
        im.append(z.imag)
    chi = devrng.device_logamp_normals(seed, 2 * real0, 2 * n) * np.sqrt(logamp_var)
    la = np.concatenate([chi[0::2], chi[1::2]])
    return R.detector(np.stack(re + im), W, dx, la)


def _vk_spectrum(N, dx, L0=np.inf):
    g = R.main_grid(N, dx)
    ps = R.von_karman(g.fabs, np.array([3e-13, 1e-13]), L0, 1e-3).sum(0) * 2 * np.pi * (2 * np.pi / 1550e-9) ** 2
    return ps, g.df


def _window_W(Np, seed=0):
    y, x = np.mgrid[0:Np, 0:Np]
    c = (Np - 1) / 2
    rr = np.hypot(x - c, y - c)
    return np.where(rr <= Np / 2 - 1, np.exp(-(rr / (0.45 * Np)) ** 2), 0.0)


# ------------------------------------------------------------------ power spectrum kernel
def _ps_call(g, p):
    prob = fast_amd.host.build_problem(fast_amd.conf.ConfigParser(dict(p)).config)
    atm = prob.atm
    return prob, _lib.powerspec(prob.N, prob.dx, prob.wvl, p["L0"], p["l0"], prob.ao_mode, p["ALIAS"], p["NOISE"],
                                prob.d_wfs, p["TLOOP"], p["TEXP"], atm.dtheta, atm.cn2, atm.h, atm.wind_vector,
                                prob.pup.pupil_filter, prob.simpson_w, lf_mask=None, modal=prob.modal,
                                modal_mult=prob.modal_mult, zmax=prob.zmax, D_ground=p["D_GROUND"],
                                per_layer=True, device=0)


# ------------------------------------------------------------------ device-RNG mode
def _small_problem(N=512, Np=82, prec="f64", scale=0.02):
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    h = f32_draw_handle(N, Np, prec, 0)
    h.set_spectrum(ps * scale, df)
    W = _window_W(Np)
    h.set_pupil(W, (N - Np) // 2, 0.01)
    return h, ps * scale, df, W
