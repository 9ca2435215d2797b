"""GPU parity: the HIP path (through the C-ABI) against the oracle and the golden fixtures.

Tolerances (identical coefficients): f64 pipeline rtol 1e-9 on per-iteration power and 1e-11 of
the screen's peak on phase; f32 pipeline rtol 1e-4 (atol 1e-9) on power.  Device-generator mode:
DEVICE_RTOL = 1e-5 on power, against the oracle fed with the device's own float32 draws coloured
in float32 as the kernels colour them (measured: ~1e-7), and against the oracle fed with the float64
restatement of the generator (oracle/devrng.py; the float32 hardware log2 / sqrt / sin / cos and the
float32 colouring then show: measured ~1e-7 too on few-radian screens).
"""
import numpy as np
import pytest

from conftest import E2E_CASES, load_golden, params_from_json
import fast_amd
from fast_amd import _lib
from oracle import fastref as R
from oracle import devrng

pytestmark = pytest.mark.gpu

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


# ------------------------------------------------------------------ generator
@pytest.mark.parametrize("N", [16, 33, 128, 256, 512, 1024, 2048, 4096, 200, 1000, 1500, 2000])
def test_device_generator_matches_oracle_restatement(N):
    h = _lib.Handle(N, max(1, N // 4), "f64", 0)
    for seed, g in ((1, 0), (0xDEADBEEFCAFE, 5), (7, 2 ** 33 + 3)):
        got = h.rng_coeffs(seed, g)
        want = devrng.device_coefficients(seed, g, N)
        err = np.abs(got - want)
        # float32 hardware log / sqrt / sin / cos against float64: ~1e-7 typically; the radius loses relative
        # accuracy where u -> 1 (|ln u| tiny), which 16.8 M draws at 4096^2 do reach
        assert err.max() < 1e-3 and np.quantile(err, 0.9999) < 1e-5
    la = h.rng_logamp(9, 2 ** 32 - 4, 16)
    assert np.abs(la - devrng.device_logamp_normals(9, 2 ** 32 - 4, 16)).max() < 1e-4
    big = h.rng_coeffs(3, 1)
    if N >= 512:
        assert abs(big.real.mean()) < 0.01 and abs(big.real.std() - 1) < 0.01 and abs(big.imag.std() - 1) < 0.01
        assert abs(np.mean(big.real * big.imag)) < 0.01


# ------------------------------------------------------------------ screens: golden FFT KATs (direct family)
@pytest.mark.parametrize("N", [16, 30, 33, 64, 128, 100, 150])
@pytest.mark.parametrize("prec,tol", [("f64", 1e-11), ("f32", 3e-5)])
def test_screens_match_reference_fft_kat(N, prec, tol):
    g = load_golden(f"kat_fft_N{N}")
    for Np in (N, max(1, N // 3), 5):
        lo = (N - Np) // 2
        h = _lib.Handle(N, Np, prec, 0)
        h.set_spectrum(g["powerspec"], float(g["df"]))
        h.set_pupil(np.ones((Np, Np)), lo, float(g["dx"]))
        phs = h.screens_coeffs(g["coeffs"].real, g["coeffs"].imag)
        want = g["screens"][:, lo:lo + Np, lo:lo + Np]
        assert np.abs(phs - want).max() <= tol * np.abs(g["screens"]).max()


# ------------------------------------------------------------------ screens: wave family vs numpy FFT
@pytest.mark.parametrize("N,Np", [(128, 22), (128, 128), (192, 82), (320, 33), (384, 128), (640, 82), (768, 82), (1280, 82), (1536, 101), (768, 300), (448, 82), (576, 82), (896, 82), (1152, 82), (1792, 82),
                                  (256, 82), (256, 200), (256, 256), (512, 82), (1024, 82), (2048, 82), (512, 23), (1024, 200), (512, 512), (1024, 1), (2048, 129),
                                  (512, 150), (1024, 129), (1024, 256), (1024, 257), (2048, 256), (2048, 300), (4096, 82), (4096, 200),
                                  (1024, 402), (1024, 512), (2048, 402), (2048, 512)])
@pytest.mark.parametrize("prec,tol", [("f64", 1e-11), ("f32", 5e-5)])
def test_wave_kernels_match_oracle_fft(N, Np, prec, tol):
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    rng = np.random.default_rng(N + Np)
    B = 2 if N < 2048 else 1
    if N == 4096 and prec == "f32":
        pytest.skip("4096^2 oracle transform once is enough")
    cr, ci = rng.normal(size=(B, N, N)), rng.normal(size=(B, N, N))
    lo = (N - Np) // 2
    want = R.crop(R.double_screens(R.screens_fftw((cr + 1j * ci) * np.sqrt(ps), df)), N, Np)
    h = _lib.Handle(N, Np, prec, 0)
    assert h.kernel_path() == 1
    h.set_spectrum(ps, df)
    h.set_pupil(np.ones((Np, Np)), lo, 0.01)
    got = h.screens_coeffs(cr, ci)
    assert np.abs(got - want).max() <= tol * np.abs(want).max()
    # the direct family on the same problem
    h.kernel_path(0)
    got_d = h.screens_coeffs(cr[:1], ci[:1])
    assert np.abs(got_d[0] - want[0]).max() <= tol * np.abs(want).max()
    assert np.abs(got_d[1] - want[B]).max() <= tol * np.abs(want).max()


def test_wave_window_not_centred():
    N, Np = 512, 70
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    rng = np.random.default_rng(5)
    cr, ci = rng.normal(size=(1, N, N)), rng.normal(size=(1, N, N))
    full = R.double_screens(R.screens_fftw((cr + 1j * ci) * np.sqrt(ps), df))
    for lo in (0, 3, N - Np):
        h = _lib.Handle(N, Np, "f64", 0)
        h.set_spectrum(ps, df)
        h.set_pupil(np.ones((Np, Np)), lo, 0.01)
        got = h.screens_coeffs(cr, ci)
        assert np.abs(got - full[:, lo:lo + Np, lo:lo + Np]).max() <= 1e-11 * np.abs(full).max()


# ------------------------------------------------------------------ detector + log-amplitude
@pytest.mark.parametrize("N,Np", [(64, 22), (512, 82)])
@pytest.mark.parametrize("prec,rtol", [("f64", 1e-9), ("f32", 1e-4)])
@pytest.mark.parametrize("coherent", [False, True])
def test_powers_match_oracle(N, Np, prec, rtol, coherent):
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    ps = ps * 0.02   # ~ few rad rms so that powers are not all tiny
    rng = np.random.default_rng(3)
    B = 3
    cr, ci = rng.normal(size=(B, N, N)), rng.normal(size=(B, N, N))
    la = rng.normal(scale=0.1, size=2 * B)
    W = _window_W(Np)
    want = R.powers_from_coefficients(cr + 1j * ci, ps, df, W, 0.01, la, coherent)
    h = _lib.Handle(N, Np, prec, 0)
    h.set_spectrum(ps, df)
    h.set_pupil(W, (N - Np) // 2, 0.01)
    got = h.run_coeffs(cr, ci, la, coherent)
    if coherent:
        assert got.dtype == complex
        assert np.abs(got - want).max() <= rtol * np.abs(want).max() * 10
    else:
        np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-9)


def test_detector_kat_from_reference():
    """The reference's Fast.compute_detector (fast/fast.py:647-668) on an explicit phase cube: the fixture's `phs`
    goes through the DEVICE detector and must give its `incoherent` and `coherent`.  The four 22 x 22 phase planes
    are tiled into one 64 x 64 layer screen and sampled by the frozen-flow entry point at integer coordinates
    (bilinear weights 0 and 1: the plane itself), so the W exp(i phi) reduction, the normalisation and exp(chi) that
    run are the kernel's own (k_temporal_detect; the screen path's epilogue is pinned through `_r` of the e2e fixtures)."""
    g = load_golden("kat_detector")
    phs, W, M = g["phs"], g["W"], int(g["M"])
    Np, N = W.shape[0], 64
    la = g["logamp"][int(g["chunk"]) * M:(int(g["chunk"]) + 1) * M]
    screen = np.zeros((1, N, N))
    corners = [(0, 0), (0, Np + 3), (Np + 5, 1), (Np + 7, Np + 9)]
    for j, (r0, c0) in enumerate(corners):
        screen[0, r0:r0 + Np, c0:c0 + Np] = phs[j]
    xs = np.stack([r0 + np.arange(Np, dtype=float) for r0, _ in corners])[None]       # (L=1, M, Np) rows
    ys = np.stack([c0 + np.arange(Np, dtype=float) for _, c0 in corners])[None]
    h = _lib.Handle(N, Np, "f64", 0)
    h.set_pupil(W, (N - Np) // 2, float(g["dx"]))
    h.set_layer_screens(screen)
    inc = h.temporal_chunk(xs, ys, np.zeros((1, 2, M), dtype=np.int32), la, coherent=False)
    coh = h.temporal_chunk(xs, ys, np.zeros((1, 2, M), dtype=np.int32), la, coherent=True)
    np.testing.assert_allclose(inc, g["incoherent"], rtol=1e-12)
    np.testing.assert_allclose(coh, g["coherent"], rtol=1e-12, atol=1e-15)
    # and the screen path's epilogue on a zero spectrum: phi = 0 -> power = exp(2 chi)
    h.set_spectrum(np.zeros((N, N)), 1.0)
    got = h.run_coeffs(np.ones((2, N, N)), np.ones((2, N, N)), la)
    np.testing.assert_allclose(got, np.exp(2 * la), rtol=1e-12)


# ------------------------------------------------------------------ power spectrum kernel
def _ps_call(g, p):
    prob = fast_amd.host.build_problem(fast_amd.conf.ConfigParser(dict(p)).config)
    atm = prob.atm
    return prob, _lib.powerspec(prob.N, prob.dx, prob.wvl, p["L0"], p["l0"], prob.ao_mode, p["ALIAS"], p["NOISE"],
                                prob.d_wfs, p["TLOOP"], p["TEXP"], atm.dtheta, atm.cn2, atm.h, atm.wind_vector,
                                prob.pup.pupil_filter, prob.simpson_w, lf_mask=None, modal=prob.modal,
                                modal_mult=prob.modal_mult, zmax=prob.zmax, D_ground=p["D_GROUND"],
                                per_layer=True, device=0)


@pytest.mark.parametrize("case", E2E_CASES + ["default164"])
def test_powerspec_kernel_matches_reference(case):
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    prob, out = _ps_call(g, p)
    peak = np.abs(g["powerspec"]).max()
    np.testing.assert_allclose(out["lf_mask"], g["lf_mask"], rtol=1e-11, atol=1e-14)     # mask_lf on the device
    np.testing.assert_allclose(out["powerspec"], g["powerspec"], rtol=1e-10, atol=1e-13 * peak)
    np.testing.assert_allclose(out["powerspec_per_layer"], g["powerspec_per_layer"], rtol=1e-10, atol=1e-13 * peak)
    np.testing.assert_allclose(out["logamp_powerspec"], g["logamp_powerspec"], rtol=1e-10,
                               atol=1e-13 * np.abs(g["logamp_powerspec"]).max())
    for k in ("logamp_var", "phs_var", "fitting_error", "aniso_servo_error", "alias_error", "noise_error"):
        np.testing.assert_allclose(out[k], g[k], rtol=1e-9, atol=1e-300, err_msg=k)
    np.testing.assert_allclose(out["phs_var_weights"], g["phs_var_weights"], rtol=1e-9)


@pytest.mark.parametrize("name", ["big_noao_1024", "big_noao_L0_1024", "big_ao_1024", "cfg1_256", "big_noao_L0_2048", "big_noao_L0_4096",
                                  "big_tt_1024", "big_lgsao_1024", "big_modal_zmax_noise_1024", "big_subharm_coherent_down_1024",
                                  "big_zenith05_1024", "big_zenith27_1024", "big_ao_1000"])
def test_powerspec_kernel_full_size(name):
    g = load_golden(name)
    p = params_from_json(g["params_json"])
    prob, out = _ps_call(g, p)
    s = int(g["stride"])
    N = prob.N
    peak = np.abs(g["powerspec_centre"]).max()
    np.testing.assert_allclose(out["powerspec"][::s, ::s], g["powerspec_strided"], rtol=1e-10, atol=1e-13 * peak)
    np.testing.assert_allclose(out["powerspec"][N // 2 - 16:N // 2 + 16, N // 2 - 16:N // 2 + 16], g["powerspec_centre"],
                               rtol=1e-10, atol=1e-13 * peak)
    np.testing.assert_allclose(out["powerspec"].sum(), g["powerspec_sum"], rtol=1e-10)
    np.testing.assert_allclose(out["lf_mask"].sum(), g["lf_mask_sum"], rtol=1e-12)
    for k in ("logamp_var", "phs_var", "fitting_error", "aniso_servo_error", "alias_error"):
        np.testing.assert_allclose(out[k], g[k], rtol=1e-9, atol=1e-300, err_msg=k)


# ------------------------------------------------------------------ end to end: Fast(config).run()
@pytest.mark.parametrize("case", E2E_CASES + ["default164"])
def test_fast_run_reproduces_reference_same_seed(case):
    """GPU_RNG='host': numpy draws in the reference's order -> the reference's result._r."""
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "host", "GPU_DEVICE": 0})
    sim = fast_amd.Fast(p)
    res = sim.run()
    assert res._r.dtype == g["r"].dtype
    np.testing.assert_allclose(res._r, g["r"], rtol=1e-9)
    np.testing.assert_allclose(sim.logamp, g["logamp"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(sim.diffraction_limit, g["diffraction_limit"], rtol=1e-12)
    assert np.isfinite(res.power).all() and np.isfinite(res.dB_rel).all() and np.isfinite(res.dB_abs).all()
    if "phs_last_chunk" in g.files and case != "numpy_branch":
        # Fast.phs after run() = the last chunk's screens (fast.py:596-603), also in host-generator mode
        np.testing.assert_allclose(sim.phs, g["phs_last_chunk"], rtol=1e-9, atol=1e-11 * np.abs(g["phs_last_chunk"]).max())


@pytest.mark.parametrize("case", ["ao_alias", "noao_L0", "subharm"])
def test_fast_run_f32_close_to_reference(case):
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "host", "GPU_DEVICE": 0, "GPU_PRECISION": "f32"})
    np.testing.assert_allclose(fast_amd.Fast(p).run()._r, g["r"], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("name", ["cfg1_256", "big_noao_1024", "big_noao_L0_1024", "big_ao_1024", "big_noao_L0_2048", "big_noao_L0_4096",
                                  "big_tt_1024", "big_lgsao_1024", "big_modal_zmax_noise_1024", "big_subharm_coherent_down_1024",
                                  "big_zenith05_1024", "big_zenith27_1024", "big_ao_1000", "big_noao_1024_s1", "big_noao_1024_s2",
                                  "big_noao_L0_1024_s1", "big_noao_L0_1024_s2"])
def test_fast_run_full_size_same_seed(name):
    g = load_golden(name)
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "host", "GPU_DEVICE": 0})
    sim = fast_amd.Fast(p)
    np.testing.assert_allclose(sim._prob.W, g["W"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(sim.run()._r, g["r"], rtol=1e-8)


# ------------------------------------------------------------------ device-RNG mode
def _small_problem(N=512, Np=82, prec="f64", scale=0.02):
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    h = _lib.Handle(N, Np, prec, 0)
    h.set_spectrum(ps * scale, df)
    W = _window_W(Np)
    h.set_pupil(W, (N - Np) // 2, 0.01)
    return h, ps * scale, df, W


@pytest.mark.parametrize("N", [64, 128, 256, 512, 1000, 1024, 2048, 4096])
def test_device_rng_run_matches_oracle_with_restated_generator(N):
    """2048 and 4096 run as 2 resp. 4 interleaved sub-rows of 1024 (split wave kernels), with 128 resp. 256
    generator streams per row; 128 / 256 / 512 as packed rows with 8 / 16 / 32 streams per row; the device result must
    follow the restated generator there too."""
    Np = 22 if N == 64 else 82
    h, ps, df, W = _small_problem(N, Np)
    seed, real0, n = 42, 5, (4 if N <= 512 else 2)
    got = h.run(seed, real0, n, None, 0.01)
    coeffs = np.stack([devrng.device_coefficients(seed, real0 + j, N) for j in range(n)])
    it = 2 * real0
    chi = devrng.device_logamp_normals(seed, it, 2 * n) * 0.1
    la = np.concatenate([chi[0::2], chi[1::2]])
    want = R.powers_from_coefficients(coeffs, ps, df, W, 0.01, la)
    np.testing.assert_allclose(got, want, rtol=DEVICE_RTOL)
    if N in (128, 256, 512, 2048):
        h.kernel_path(0)                               # the direct family draws the same streams
        np.testing.assert_allclose(h.run(seed, real0, n, None, 0.01), got, rtol=1e-9)


def test_device_rng_invariant_to_batch_and_split():
    h, ps, df, W = _small_problem()
    ref = h.run(7, 0, 24, None, 0.02)
    h.set_batch(5)
    np.testing.assert_array_equal(h.run(7, 0, 24, None, 0.02), ref)
    h.set_batch(0)
    a = h.run(7, 0, 10, None, 0.02)
    b = h.run(7, 10, 14, None, 0.02)
    np.testing.assert_array_equal(np.r_[a[:10], b[:14], a[10:], b[14:]], ref)
    h.kernel_path(0)   # direct family: same generator, same answers to rounding
    np.testing.assert_allclose(h.run(7, 0, 4, None, 0.02), np.r_[ref[:4], ref[24:28]], rtol=1e-9)


def test_device_rng_statistics_match_host_mode():
    """Same distribution as numpy-drawn coefficients: mean dB within 3 sigma, KS p > 0.01."""
    from scipy import stats
    N, Np, n = 512, 82, 600
    h, ps, df, W = _small_problem(N, Np, "f32")
    dev = h.run(123, 0, n, None, 0.01)
    rng = np.random.default_rng(0)
    host = []
    for _ in range(n // 50):
        cr, ci = rng.normal(size=(50, N, N)), rng.normal(size=(50, N, N))
        host.append(h.run_coeffs(cr, ci, rng.normal(scale=0.1, size=100)))
    host = np.concatenate(host)
    d1, d2 = 10 * np.log10(dev), 10 * np.log10(host)
    se = np.sqrt(d1.var() / len(d1) + d2.var() / len(d2))
    assert abs(d1.mean() - d2.mean()) < 4 * se
    assert stats.ks_2samp(d1, d2).pvalue > 0.01
    si1, si2 = (dev / dev.mean()).var(), (host / host.mean()).var()
    assert abs(si1 / si2 - 1) < 0.25


def test_fast_device_mode_end_to_end():
    g = load_golden("e2e_ao_alias")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "NITER": 400, "NCHUNKS": 4, "SEED": 5})
    sim = fast_amd.Fast(dict(p))
    r1 = sim.run()._r
    assert r1.shape == (400,) and np.isfinite(r1).all() and (r1 > 0).all()
    p2 = dict(p)
    p2["NCHUNKS"] = 2
    r2 = fast_amd.Fast(p2).run()._r   # chunking only reorders [Re block | Im block] per chunk
    np.testing.assert_allclose(np.sort(r1), np.sort(r2), rtol=0, atol=0)
    ref = g["r"]
    assert abs(10 * np.log10(r1.mean()) - 10 * np.log10(ref.mean())) < 1.0


# ------------------------------------------------------------------ properties at BASELINE size
def test_full_size_properties_1024():
    N, Np = 1024, 82
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    W = _window_W(Np)
    h = _lib.Handle(N, Np, "f64", 0)
    h.set_pupil(W, (N - Np) // 2, 0.01)
    # zero spectrum: every power is exactly exp(2 chi)
    h.set_spectrum(np.zeros((N, N)), df)
    la = np.linspace(-0.3, 0.3, 8)
    out = h.run(1, 0, 4, la, 0.0)
    np.testing.assert_allclose(out, np.exp(2 * la), rtol=1e-12)
    # linearity of the screens in the coefficients and in sqrt(powerspec)
    h.set_spectrum(ps, df)
    a = h.screens(9, 0, 1)
    h.set_spectrum(4 * ps, df)
    b = h.screens(9, 0, 1)
    np.testing.assert_allclose(b, 2 * a, rtol=1e-12, atol=1e-12 * np.abs(a).max())
    # Parseval-type check: in-window variance over many screens ~ integral of the PSD over the grid
    h.set_spectrum(ps, df)
    scr = h.screens(3, 0, 64)
    expect = (ps * df ** 2).sum()
    assert abs(scr.var() / expect - 1) < 0.5   # piston-dominated, large sample variance
    # coherent vs incoherent consistency
    inc = h.run(11, 0, 6, None, 0.01)
    coh = h.run(11, 0, 6, None, 0.01, coherent=True)
    np.testing.assert_allclose(np.abs(coh) ** 2, inc, rtol=1e-12)


def test_histogram_matches_numpy():
    h, ps, df, W = _small_problem()
    out = h.run(5, 0, 500, None, 0.01)
    bins = h.histogram(-30.0, 5.0, 70)
    db = 10 * np.log10(out)
    want, _ = np.histogram(db, bins=70, range=(-30.0, 5.0))
    inside = (db >= -30) & (db < 5)
    assert bins[:70].sum() == inside.sum() and bins[70] == (db < -30).sum() and bins[71] == (db >= 5).sum()
    assert np.abs(bins[:70] - want).sum() <= 2   # numpy's last bin is closed; edges may move one count


# ------------------------------------------------------------------ error behaviour
def test_errors_are_reported_not_fatal():
    with pytest.raises(fast_amd.FastMCError):
        _lib.Handle(8194, 10)
    with pytest.raises(fast_amd.FastMCError):
        _lib.Handle(4100, 300)              # beyond 4096 a window above 256 pixels needs a sub-row grid
    with pytest.raises(fast_amd.FastMCError):
        _lib.Handle(64, 65)
    h = _lib.Handle(64, 22, "f64", 0)
    with pytest.raises(fast_amd.FastMCError, match="set_spectrum"):
        h.run(1, 0, 2)
    with pytest.raises(fast_amd.FastMCError):
        h.set_pupil(np.ones((22, 22)), 60, 0.01)
    with pytest.raises(fast_amd.FastMCError):
        h.set_spectrum(-np.ones((64, 64)), 1.0)
    with pytest.raises(Exception, match="NCHUNKS must divide"):
        fast_amd.Fast({"NITER": 10, "NCHUNKS": 3, "LOGLEVEL": "ERROR"})


def test_rccl_exchange_world_of_one():
    """The in-library RCCL path (dlopen, communicator, all-gather, all-reduce) with one rank."""
    h, ps, df, W = _small_problem()
    out = h.run(5, 0, 100, None, 0.01)
    h.comm_init(_lib.comm_unique_id(), 1, 0)
    allp, hist = h.comm_gather(200, 1, (-30.0, 5.0, 70))
    np.testing.assert_array_equal(allp, out)
    np.testing.assert_array_equal(hist, h.histogram(-30.0, 5.0, 70))
    assert hist.sum() == 200


def test_mask_kats_on_device():
    """ao_power_spectra.mask_lf variants (kat_masks) evaluated by the power-spectrum kernel, and the
    host-supplied-mask path (mask_mode 0) giving the same spectrum."""
    g = load_golden("kat_masks")
    N, dx = int(g["N"]), float(g["dx"])
    w = fast_amd.hostmath.simpson_weights(fast_amd.host.freq_axis(N, dx))
    common = dict(N=N, dx=dx, wvl=1550e-9, L0=25.0, l0=0.01, ao_mode="AO", alias=True, noise=0.2, d_wfs=0.08,
                  t_loop=1e-3, t_exp=1e-3, dtheta=[4, 0], cn2=np.array([1e-13, 2e-14]), h=np.array([1e3, 8e3]),
                  wind=np.array([[5.0, 0.0], [0.0, 20.0]]), pupil_filter=None, simpson_w=w, device=0)
    for key, kw in (("zonal", {}), ("modal", dict(modal=True, modal_mult=0.7)),
                    ("zern3", dict(modal=True, zmax=3, D_ground=0.4)), ("zern9", dict(modal=True, zmax=9, D_ground=0.4))):
        out = _lib.powerspec(lf_mask=None, **kw, **common)
        np.testing.assert_allclose(out["lf_mask"], g[key], rtol=1e-11, atol=1e-14, err_msg=key)
        again = _lib.powerspec(lf_mask=np.asarray(g[key], dtype=float), **common)
        np.testing.assert_allclose(again["powerspec"], out["powerspec"], rtol=1e-12)


def test_zenith_scan_sweep():
    """BASELINE config 5 pattern at reduced size: per-angle Fast objects, GPU power spectrum each."""
    from fast_amd import sweep
    g = load_golden("e2e_ao_alias")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "NPXLS": 512, "SEED": 3})
    angles = np.linspace(0, 60, 4)
    recs = sweep.gather_records(sweep.zenith_scan(p, angles, niter=64))
    assert [r["index"] for r in recs] == [0, 1, 2, 3]
    assert all(np.isfinite(r["mean_dB_rel"]) for r in recs)
    # more air mass -> smaller r0 along the line of sight, larger residual phase variance
    assert recs[0]["r0_los"] > recs[-1]["r0_los"] and recs[0]["phs_var"] < recs[-1]["phs_var"]
    half = sweep.zenith_scan(p, angles, niter=64, rank=1, world=2)
    assert [r["index"] for r in half] == [1, 3]
    assert half[0]["mean_dB_rel"] == recs[1]["mean_dB_rel"]


def test_config5_zenith_scan_full_size():
    """BASELINE configs[4] at size: 32 zenith angles x 4096 iterations at 1024^2, AO + alias, through
    sweep.zenith_scan (one Fast object per angle, spectrum evaluated and kept on the GPU).  Two of the angles are
    pinned to the reference (fixtures big_zenith05 / big_zenith27: same parameters, captured by
    tools/capture_golden/capture.py:zenith): the Simpson scalars of the device spectrum to 1e-9, and
    test_fast_run_full_size_same_seed reproduces their `_r`; the rest must follow the physics monotonically."""
    import time
    from fast_amd import sweep
    g5, g27 = load_golden("big_zenith05_1024"), load_golden("big_zenith27_1024")
    base = params_from_json(g5["params_json"])
    for k in ("ZENITH_ANGLE", "NITER", "NCHUNKS"):
        base.pop(k)
    base.update({"GPU_DEVICE": 0, "SEED": 1, "GPU_RNG": "device"})
    angles = np.linspace(0, 70, 32)
    assert angles[5] == params_from_json(g5["params_json"])["ZENITH_ANGLE"] and angles[27] == params_from_json(g27["params_json"])["ZENITH_ANGLE"]
    t0 = time.perf_counter()
    recs = sweep.zenith_scan(base, angles, niter=4096, keep_power=True)
    wall = time.perf_counter() - t0
    assert len(recs) == 32 and all(r["r"].shape == (4096,) and np.isfinite(r["r"]).all() and (r["r"] > 0).all() for r in recs)
    for idx, g in ((5, g5), (27, g27)):
        for k in ("phs_var", "logamp_var", "r0_los", "L"):
            np.testing.assert_allclose(recs[idx][k], g[k], rtol=1e-9, err_msg=f"{k} at angle {idx}")
        # 4096 device-generator iterations against the reference's 8 numpy-seeded ones: same distribution
        z = (np.log(g["r"]).mean() - np.log(recs[idx]["r"]).mean()) / (np.log(recs[idx]["r"]).std() / np.sqrt(8))
        assert abs(z) < 5
    r0 = np.array([r["r0_los"] for r in recs])
    pv = np.array([r["phs_var"] for r in recs])
    lv = np.array([r["logamp_var"] for r in recs])
    mean_db = np.array([r["mean_dB_rel"] for r in recs])
    assert (np.diff(r0) < 0).all() and (np.diff(pv) > 0).all() and (np.diff(lv) > 0).all()     # more air mass, every step
    assert mean_db[0] > mean_db[-1] + 3 and np.corrcoef(mean_db, pv)[0, 1] < -0.9
    assert recs[0]["scintillation_index"] < recs[-1]["scintillation_index"]
    assert wall < 5.0, wall                    # seconds; the reference needs 32 x (12 s init + 5 min run)


def test_config4_full_size_two_handles():
    """BASELINE configs[3] at size: 2048^2, 100 000 iterations, split over two handles (two worker threads; on a 1-GPU box
    both on device 0): the assembled vector is bit-identical to the unsharded run and the dB histogram counts every
    iteration."""
    g = load_golden("big_noao_L0_2048")
    p = params_from_json(g["params_json"])
    p.update({"NITER": 100000, "NCHUNKS": 100, "SEED": 9, "GPU_RNG": "device"})
    one = fast_amd.Fast(dict(p, GPU_DEVICE=0))
    want = one.run()._r
    assert want.shape == (100000,) and np.isfinite(want).all() and (want > 0).all()
    two = fast_amd.Fast(dict(p, GPU_DEVICES=[0, 0]))
    got = two.run()._r
    assert two._group.world == 2 and np.array_equal(got, want)
    hist = two.histogram(-60.0, 10.0, 4096)
    assert hist.sum() == 100000 and np.array_equal(hist, one.histogram(-60.0, 10.0, 4096))
    # the reference's own 4 iterations of this configuration lie inside the distribution
    lo, hi = np.quantile(want, [0.001, 0.999])
    assert ((g["r"] > lo / 3) & (g["r"] < hi * 3)).all()


def test_config4_geometry_2048():
    """2048^2 grid (BASELINE config 4 geometry) through Fast: device RNG, finite results, and the
    same statistics as the 1024^2 run of the same physical problem within sampling error."""
    g = load_golden("big_noao_L0_1024")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "NPXLS": 2048, "NITER": 512, "NCHUNKS": 4, "SEED": 9, "GPU_RNG": "device"})
    sim = fast_amd.Fast(p)
    assert sim._handle.kernel_path() == 1
    r = sim.run()._r
    assert r.shape == (512,) and np.isfinite(r).all() and (r > 0).all()
    hist = sim.histogram(-60.0, 10.0, 4096)
    assert hist.sum() == 512


def test_many_realisations_cross_finalize_span():
    """More than 32768 realisations in one call: detector partials are finalised in several spans."""
    N, Np = 64, 10
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    h = _lib.Handle(N, Np, "f32", 0)
    h.set_spectrum(ps * 0.02, df)
    h.set_pupil(_window_W(Np), (N - Np) // 2, 0.01)
    n = 33000
    full = h.run(3, 0, n, None, 0.01)
    assert np.isfinite(full).all() and (full > 0).all()
    a = h.run(3, 0, 20000, None, 0.01)
    b = h.run(3, 20000, 13000, None, 0.01)
    np.testing.assert_array_equal(full, np.r_[a[:20000], b[:13000], a[20000:], b[13000:]])
    one = h.run(3, 32999, 1, None, 0.01)
    np.testing.assert_array_equal(one, [full[32999], full[n + 32999]])


@pytest.mark.parametrize("name", ["temporal_default", "temporal_small", "temporal_noao", "temporal_npxls100"])
def test_temporal_mode_reproduces_reference(name):
    """TEMPORAL (frozen-flow) runs, incl. the reference's shipped test_params.py: same SEED -> same series."""
    g = load_golden(name)
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0})
    sim = fast_amd.Fast(p)
    np.testing.assert_allclose(sim.temporal_logamp_powerspec, g["temporal_logamp_powerspec"], rtol=1e-9, atol=1e-30)
    np.testing.assert_allclose(sim.pixel_shifts, g["pixel_shifts"], rtol=1e-13)
    res = sim.run()
    assert res._r.dtype == g["r"].dtype
    np.testing.assert_allclose(sim.logamp, g["logamp"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(res._r, g["r"], rtol=1e-8)
    # Fast.phs after a TEMPORAL run = the last chunk's summed, shifted layer phases (fast.py:619-633)
    np.testing.assert_allclose(sim.phs, g["phs_last_chunk"], rtol=1e-8, atol=1e-10 * np.abs(g["phs_last_chunk"]).max())
    r2 = sim.run()._r                   # a second run of the same object continues the generator stream, like the reference
    assert r2.shape == res._r.shape and np.isfinite(r2).all()


def test_device_generator_statistical_quality():
    """Moments, tails, uniformity of phase and independence across lanes / rows / realisations of the
    device generator (Philox-seeded xoshiro128+ streams + hardware Box-Muller), 4 x 1024^2 draws."""
    from scipy import stats
    h = _lib.Handle(1024, 8, "f64", 0)
    c = np.stack([h.rng_coeffs(2024, g) for g in range(4)])          # (4, 1024, 1024) complex
    z = np.concatenate([c.real.ravel(), c.imag.ravel()])
    n = z.size
    assert abs(z.mean()) < 5 / np.sqrt(n)
    assert abs(z.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs(stats.skew(z)) < 5 * np.sqrt(6 / n)
    assert abs(stats.kurtosis(z)) < 5 * np.sqrt(24 / n)
    # tails against the normal law: expected counts beyond 3, 4, 5 sigma
    for k in (3.0, 4.0, 5.0):
        expect = n * 2 * stats.norm.sf(k)
        got = np.count_nonzero(np.abs(z) > k)
        assert abs(got - expect) < 6 * np.sqrt(expect) + 3
    # |c|^2 / 2 is Exp(1), the phase is uniform
    assert stats.kstest((np.abs(c[0]) ** 2 / 2).ravel()[::7], "expon").pvalue > 1e-3
    assert stats.kstest((np.angle(c[1]).ravel()[::7] + np.pi) / (2 * np.pi), "uniform").pvalue > 1e-3
    # independence: neighbouring lanes, neighbouring stream positions (kx, kx+64), rows, realisations
    def corr(a, b):
        return abs(np.mean(a * b)) * np.sqrt(a.size)
    re = c.real
    assert corr(re[:, :, :-1], re[:, :, 1:]) < 5
    assert corr(re[:, :, :-64], re[:, :, 64:]) < 5
    assert corr(re[:, :-1, :], re[:, 1:, :]) < 5
    assert corr(re[0], re[1]) < 5 and corr(re[0], c.imag[0]) < 5
    # different seeds decorrelate
    assert corr(re[0], h.rng_coeffs(2025, 0).real) < 5


def test_result_stats_on_device_match_numpy():
    h, ps, df, W = _small_problem()
    out = h.run(5, 0, 3000, None, 0.01)
    thr = [10 ** (-d / 10) for d in (3.0, 6.0, 10.0)]
    st = h.result_stats(thr)
    res = fast_amd.FastResult(out, 1.0)
    assert st["n"] == 6000
    np.testing.assert_allclose(st["mean"], out.mean(), rtol=1e-12)
    np.testing.assert_allclose(st["scintillation_index"], res.scintillation_index, rtol=1e-9)
    np.testing.assert_allclose(st["avg_dB_rel"], res.avg_power_dB_rel, rtol=1e-12)
    np.testing.assert_allclose(st["mean_dB_rel"], res.dB_rel.mean(), rtol=1e-12)
    assert st["min"] == out.min() and st["max"] == out.max()
    np.testing.assert_array_equal(st["fade_prob"], [(out < t).mean() for t in thr])
    coh = h.run(5, 0, 100, None, 0.01, coherent=True)
    st2 = h.result_stats()
    np.testing.assert_allclose(st2["mean"], (np.abs(coh) ** 2).mean(), rtol=1e-12)


def test_fast_object_attributes_like_the_reference():
    """freq grids and the last chunk's phase screens (fast.py:49-64, 596-603, 814-875)."""
    g = load_golden("e2e_subharm_ao")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "NITER": 40, "NCHUNKS": 4})
    sim = fast_amd.Fast(p)
    grid = R.main_grid(sim.Npxls, sim.dx)
    np.testing.assert_allclose(sim.freq.main.fabs, grid.fabs, rtol=1e-15)
    np.testing.assert_allclose(sim.freq.fx, grid.fx, rtol=1e-15)
    np.testing.assert_allclose(sim.freq.subharm.fx, g["sh_fx"], rtol=1e-15)
    assert sim.freq.df == grid.df and sim.freq.main.f.shape == (sim.Npxls,)
    res = sim.run()
    phs = sim.phs
    assert phs.shape == (10, sim.Npxls_pup, sim.Npxls_pup)
    # detector of those screens == the last chunk's results (oracle formula on GPU screens)
    want = R.detector(phs, sim.pupil * sim.pupil_mode, sim.dx, sim.logamp[-10:])
    np.testing.assert_allclose(res._r[-10:], want, rtol=1e-9)


@pytest.mark.parametrize("key,pkey", [("r_ao", "params_json"), ("r_noao", "params2_json")])
def test_device_mode_distribution_matches_reference_output(key, pkey):
    """4000 GPU iterations with the device generator vs 2000 iterations of the REFERENCE itself
    (numpy PCG64 draws) on the same configuration: same distribution (two-sample KS), same mean
    power within 4 standard errors, same scintillation index within 25 %."""
    from scipy import stats
    g = load_golden("stat_ref_256")
    ref = g[key]
    p = params_from_json(g[pkey])
    p.update({"GPU_DEVICE": 0, "NITER": 4000, "NCHUNKS": 20, "SEED": 1234, "GPU_RNG": "device"})
    r = fast_amd.Fast(p).run()._r
    d1, d2 = 10 * np.log10(r), 10 * np.log10(ref)
    assert stats.ks_2samp(d1, d2).pvalue > 0.01
    se = np.sqrt(r.var() / r.size + ref.var() / ref.size)
    assert abs(r.mean() - ref.mean()) < 4 * se
    si1, si2 = (r / r.mean()).var(), (ref / ref.mean()).var()
    assert abs(si1 / si2 - 1) < 0.25


@pytest.mark.parametrize("N", [256, 1024, 2048])
def test_f32_pipeline_tracks_f64_on_the_same_device_draws(N):
    """Same seed -> same generator words in both precisions; only the transform arithmetic differs.
    Full-size check of the float32 pipeline against the float64 one (phases of tens of radians)."""
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    W = _window_W(82)
    out = {}
    for prec in ("f64", "f32"):
        h = _lib.Handle(N, 82, prec, 0)
        h.set_spectrum(ps, df)
        h.set_pupil(W, (N - 82) // 2, 0.01)
        out[prec] = h.run(77, 3, 16, None, 0.01)
        scr = h.screens(77, 3, 1)
        out[prec + "_rms"] = scr.std()
    assert out["f64_rms"] > 0.5                      # radians rms over the window of one screen (piston-dominated: varies per draw)
    np.testing.assert_allclose(out["f32"], out["f64"], rtol=5e-4, atol=1e-9)


def test_link_metrics_match_reference_fixtures():
    """fast_amd.comms (device reductions) vs the reference's fast/comms.py:171-262 outputs: integer counts
    behind fade_prob / fade_dur exact (NaN conventions included); erfc integrals rtol 1e-11."""
    from fast_amd import comms
    d = load_golden("comms_metrics")
    thr, eb, Ms, dt = d["thresholds"], d["ebn0"], d["Ms"], float(d["dt"])
    for n in d["names"]:
        v = d["v_" + n]
        np.testing.assert_array_equal([comms.fade_prob(v, t) for t in thr], d["fade_prob_" + n])
        np.testing.assert_array_equal([comms.fade_prob(v, t, 5) for t in thr], d["fade_prob_min5_" + n])
        np.testing.assert_allclose([comms.fade_dur(v, t, dt) for t in thr], d["fade_dur_" + n], rtol=1e-14)
        np.testing.assert_allclose([comms.fade_dur(v, t, dt, 5) for t in thr], d["fade_dur_min5_" + n], rtol=1e-14)
        np.testing.assert_allclose([comms.ber_ook(s, v) for s in eb], d["ber_ook_" + n], rtol=1e-11)
        np.testing.assert_allclose([[comms.sep_qam(M, s, v) for s in eb] for M in Ms], d["sep_qam_" + n], rtol=1e-11)
        np.testing.assert_allclose([[comms.ber_qam(M, s, v) for s in eb] for M in Ms], d["ber_qam_" + n], rtol=1e-11)
    np.testing.assert_allclose([comms.ber_ook(s) for s in eb], d["ber_ook_nosamples"], rtol=1e-12)
    np.testing.assert_allclose([[comms.ber_qam(M, s) for s in eb] for M in Ms], d["ber_qam_nosamples"], rtol=1e-12)
    np.testing.assert_allclose(comms.Q(np.array([-1.0, 0.0, 0.5, 3.0])), R.q_function(np.array([-1.0, 0.0, 0.5, 3.0])), rtol=1e-12)


def test_link_metrics_on_resident_results_match_oracle():
    """Metrics reduced where the run left its results (no vector transfer) equal the oracle's on the
    returned vector; random long series incl. fades crossing block boundaries vs the run-length oracle."""
    from fast_amd import comms
    p = params_from_json(load_golden("e2e_noao_L0")["params_json"])
    p.update({"NITER": 4000, "NCHUNKS": 4, "SEED": 5, "GPU_RNG": "device"})
    sim = fast_amd.Fast(p)
    r = sim.run()._r
    thr = float(np.quantile(r, 0.2))
    assert comms.fade_prob(sim, thr) == R.fade_prob(r, thr)
    # one threshold, one unit (power relative to the diffraction limit) for every function that takes the object
    assert comms.fade_prob(sim, thr) == comms.fade_prob(sim.result._r, thr)
    a, b = comms.fade_dur(sim, thr, 1e-3, 5), R.fade_dur(r, thr, 1e-3, 5)
    assert (np.isnan(a) and np.isnan(b)) or a == b
    a, b = comms.fade_dur(sim, thr, 1e-3, 5), comms.fade_dur(sim.result._r, thr, 1e-3, 5)
    assert (np.isnan(a) and np.isnan(b)) or a == b
    assert comms.fade_counts(sim, thr)[:2] == (4000, int((r < thr).sum()))
    np.testing.assert_allclose(comms.ber_ook(8.0, sim), R.ber_ook(8.0, r), rtol=1e-11)
    np.testing.assert_allclose(comms.ber_qam(16, 12.0, sim), R.ber_qam(16, 12.0, r), rtol=1e-11)
    rng = np.random.default_rng(3)
    for n in (1, 2, 255, 256, 257, 65536 + 17, 300001):
        x = np.exp(np.convolve(rng.normal(0, 1, n + 40), np.ones(41) / 6.4, mode="valid"))
        for t in (0.5, 1.0, 2.0):
            assert comms.fade_counts(x, t)[1] == int((x < t).sum())
            a, b = comms.fade_dur(x, t, 0.5, 3), R.fade_dur(x, t, 0.5, 3)
            assert (np.isnan(a) and np.isnan(b)) or a == b
    with pytest.raises(fast_amd.FastMCError):
        _lib.link_metrics([(7, 0.0, 0.0)], samples=np.ones(4))


@pytest.mark.parametrize("N", [48, 49, 64, 100, 256, 2048, 4096])
def test_centred_fft2_matches_numpy(N):
    """_lib.centred_fft2 (the row/column kernels with the window = the whole grid; wave family for
    64, 256 and 2048 (its single-pass P = 32 kernels), direct family otherwise -- at 4096 with the twiddles
    in global memory because 3 N complex exceed the LDS; odd N with numpy's asymmetric shifts) vs numpy.fft."""
    rng = np.random.default_rng(N)
    g = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    fwd = np.fft.fftshift(np.fft.fft2(np.fft.fftshift(g)))
    inv = np.fft.ifftshift(np.fft.ifft2(np.fft.ifftshift(g)))
    np.testing.assert_allclose(_lib.centred_fft2(g), fwd, rtol=0, atol=1e-11 * np.abs(fwd).max())
    np.testing.assert_allclose(_lib.centred_fft2(g, inverse=True), inv, rtol=0, atol=1e-11 * np.abs(inv).max())


def test_mean_irradiance_matches_reference():
    """Fast.compute_mean_irradiance with its transforms on the GPU vs the reference's outputs (fast.py:736-761)."""
    from fast_amd import host
    g = load_golden("mean_irradiance")
    on = host.mean_irradiance(g["powerspec"], g["W"], float(g["dx"]), float(g["df"]), float(g["diffraction_limit"]))
    np.testing.assert_allclose(on, g["onaxis"], rtol=1e-10)
    off = host.mean_irradiance(g["powerspec"], g["W"], float(g["dx"]), float(g["df"]), float(g["diffraction_limit"]), onaxis=False)
    np.testing.assert_allclose(off, g["offaxis"], rtol=1e-9, atol=1e-11 * np.abs(g["offaxis"]).max())
    on2 = host.mean_irradiance(g["powerspec2"], g["W2"], float(g["dx2"]), float(g["df2"]), float(g["diffraction_limit2"]))
    np.testing.assert_allclose(on2, g["onaxis2"], rtol=1e-10)
    np.testing.assert_allclose(on, R.mean_irradiance(g["powerspec"], g["W"], float(g["dx"]), float(g["df"]),
                                                     float(g["diffraction_limit"])), rtol=1e-11)


@pytest.mark.parametrize("name", ["e2e_ao_alias", "e2e_coherent", "temporal_small"])
def test_device_reductions_cover_multi_call_runs(name):
    """Host-coefficient chunks and TEMPORAL chunks are several library calls: the histogram, the
    statistics and the link metrics must reduce the whole FastResult, not the last chunk."""
    from fast_amd import comms
    p = params_from_json(load_golden(name)["params_json"])
    sim = fast_amd.Fast(p)
    r = np.abs(sim.run()._r) ** 2 if p.get("COHERENT") else sim.run()._r
    assert sim.Nchunks > 1
    assert sim.histogram(-80.0, 20.0, 64).sum() == len(r)
    st = sim.result_stats()
    assert st["n"] == len(r)
    np.testing.assert_allclose(st["mean"], r.mean(), rtol=1e-12)
    np.testing.assert_allclose(comms.ber_ook(6.0, sim), R.ber_ook(6.0, r), rtol=1e-11)


@pytest.mark.parametrize("rotate", [0.0, 0.3])
@pytest.mark.parametrize("N,Np", [(128, 22), (256, 82), (48, 23)])
def test_subharmonic_screens_separable_and_general_grids(N, Np, rotate):
    """fastmc_set_subharm takes arbitrary (3,3,3) frequency grids.  The reference's are 3 x 3
    meshgrids per level (fast.py:835-844) and take the column-folded 9-term path; a rotated grid
    takes the general 27-term path.  Both against the oracle's full-grid evaluation (funcs.py:225-258)."""
    from types import SimpleNamespace
    dx = 0.01
    ps, df = _vk_spectrum(N, dx, 25.0)
    g = R.subharm_grid(N, dx)
    c, s_ = np.cos(rotate), np.sin(rotate)
    grid = SimpleNamespace(fx=c * g.fx - s_ * g.fy, fy=s_ * g.fx + c * g.fy, df=g.df)
    rng = np.random.default_rng(N)
    ps_lo = rng.uniform(0.5, 2.0, size=(3, 3, 3))
    B = 2
    cr, ci = rng.normal(size=(B, N, N)), rng.normal(size=(B, N, N))
    sr, si = rng.normal(size=(B, 3, 3, 3)), rng.normal(size=(B, 3, 3, 3))
    lo = (N - Np) // 2
    want = R.crop(R.double_screens(R.screens_fftw((cr + 1j * ci) * np.sqrt(ps), df))
                  + R.subharm_screens((sr + 1j * si) * np.sqrt(ps_lo), grid, N, dx), N, Np)
    for path in ([1, 0] if N % 64 == 0 else [0]):
        h = _lib.Handle(N, Np, "f64", 0)
        h.kernel_path(path)
        h.set_spectrum(ps, df)
        h.set_pupil(np.ones((Np, Np)), lo, dx)
        h.set_subharm(ps_lo, grid.fx, grid.fy, grid.df)
        got = h.screens_coeffs(cr, ci, sr, si)
        assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()


def test_two_handles_in_two_threads():
    """Handles are independent (own stream, own buffers; ctypes releases the GIL): two threads running
    different problems concurrently get what they get alone."""
    import threading
    jobs = [(512, 82, "f64", 11), (256, 40, "f32", 12)]
    alone, together = {}, {}

    def work(store, job):
        N, Np, prec, seed = job
        h, ps, df, W = _small_problem(N, Np, prec)
        out = [h.run(seed, 7 * k, 50, None, 0.01) for k in range(6)]
        store[job] = np.concatenate(out)
        h.close()

    for j in jobs:
        work(alone, j)
    ts = [threading.Thread(target=work, args=(together, j)) for j in jobs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for j in jobs:
        np.testing.assert_array_equal(alone[j], together[j])


@pytest.mark.parametrize("N", [16, 30, 33, 64, 128, 100, 150])
def test_funcs_make_phase_fft_like_the_reference(N):
    """fast_amd.funcs.make_phase_fft (reference signature, fast/funcs.py:210-223) against the reference's own
    outputs: full N x N screens, double=True stacks [Re | Im], double=False returns Re."""
    from fast_amd import funcs
    g = load_golden(f"kat_fft_N{N}")
    rand = g["coeffs"] * np.sqrt(g["powerspec"])
    want = g["screens"]
    got = funcs.make_phase_fft(rand, float(g["df"]), fftw=True, double=True)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    single = funcs.make_phase_fft(rand, float(g["df"]))
    assert np.abs(single - want[:len(rand)]).max() <= 1e-11 * np.abs(want).max()
    assert funcs.make_phase_fft(rand[0], float(g["df"])).shape == (N, N)


def test_funcs_make_phase_subharm_like_the_reference():
    """fast_amd.funcs.make_phase_subharm (fast/funcs.py:225-258) against the reference's output (kat_subharm.npz)."""
    from types import SimpleNamespace
    from fast_amd import funcs
    g = load_golden("kat_subharm")
    freq = SimpleNamespace(subharm=SimpleNamespace(fx=g["fx"], fy=g["fy"], df=g["df"]))
    got = funcs.make_phase_subharm(g["rand"], freq, int(g["N"]), float(g["dx"]), double=True)
    assert got.shape == g["screens"].shape
    assert np.abs(got - g["screens"]).max() <= 1e-11 * np.abs(g["screens"]).max()
    np.testing.assert_array_equal(funcs.make_phase_subharm(g["rand"], freq, int(g["N"]), float(g["dx"])), got[:len(g["rand"])])


@pytest.mark.parametrize("case", E2E_CASES)
def test_psd_terms_on_the_object_match_reference(case):
    """Fast.turb_powerspec / G_ao / alias_powerspec / noise_powerspec (fast.py:448-472: funcs.turb_powerspectrum_vonKarman,
    ao_power_spectra.G_AO_PAOLA, Jol_alias_openloop, Jol_noise_openloop) from the GPU vs the reference's attributes,
    plain scalars where the reference keeps scalars."""
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    p["GPU_DEVICE"] = 0
    sim = fast_amd.Fast(p)
    for name in ("turb_powerspec", "G_ao", "alias_powerspec", "noise_powerspec"):
        want, got = g[name], getattr(sim, name)
        assert np.shape(got) == want.shape, name
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-13 * max(np.abs(want).max(), 1e-300), err_msg=name)


def test_default_fftw_false_is_warned_and_tied_to_the_numpy_branch_fixture(caplog):
    """`FFTW: False` is the reference's default (fast/conf.py:71 -> aotools ift2, funcs.py:216-218).  The GPU path
    computes the FFTW branch whatever the flag: (i) it says so, once; (ii) its result for the same SEED is the FFTW
    branch's `_r`, not the default branch's; (iii) the full-grid device screens reproduce the default branch's screens
    of the fixture through the relation pinned in tests/test_oracle_golden.py (mirror, chunk roll, (chunk / N)^2)."""
    import logging
    from fast_amd import fast as ffast, funcs as gfuncs
    g = load_golden("e2e_numpy_branch")
    p = params_from_json(g["params_json"])
    assert p["FFTW"] is False
    ffast._BRANCH_WARNED.discard(False)
    p.update({"GPU_RNG": "host", "GPU_DEVICE": 0, "LOGLEVEL": "WARNING"})
    with caplog.at_level(logging.WARNING, logger="fast_amd.fast"):
        sim = fast_amd.Fast(dict(p))
        fast_amd.Fast(dict(p))
    msgs = [r.getMessage() for r in caplog.records if "FFTW is False" in r.getMessage()]
    assert len(msgs) == 1 and "ift2" in msgs[0] and "funcs.py:212-215" in msgs[0]
    r = sim.run()._r
    np.testing.assert_allclose(r, g["r_fftw"], rtol=1e-9)
    assert np.abs(r / g["r"] - 1).max() > 1e-3
    # (iii) device transform of the last chunk's coefficients -> the default branch's window
    N, Np = int(g["Npxls"]), int(g["Npxls_pup"])
    B = p["NITER"] // p["NCHUNKS"] // 2
    rng = np.random.default_rng(p["SEED"])
    R.draw_logamp(rng, p["NITER"], float(g["logamp_var"]))
    for _ in range(p["NCHUNKS"]):
        coeffs = R.draw_coefficients(rng, (B, N, N))
    full = gfuncs.make_phase_fft(coeffs * np.sqrt(g["powerspec"]), float(g["df"]), double=True, device=0)      # (2B, N, N)
    z = full[:B] + 1j * full[B:]
    idx = (N - np.arange(N)) % N
    zn = (B / N) ** 2 * np.roll(z[:, idx][:, :, idx], -2 * (B // 2), axis=0)
    got = R.crop(R.double_screens(zn), N, Np)
    np.testing.assert_allclose(got, g["phs_last_chunk"], rtol=1e-9, atol=1e-12 * np.abs(g["phs_last_chunk"]).max())


def test_subharm_bookkeeping_attributes_like_the_reference():
    """powerspec_subharm_per_layer, phs_var_subharm, phs_var_weights_sh, lf_mask_subharm (fast.py:494-526) on the object."""
    g = load_golden("e2e_subharm_ao")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0})
    sim = fast_amd.Fast(p)
    np.testing.assert_allclose(sim.powerspec_subharm_per_layer, g["powerspec_subharm_per_layer"], rtol=1e-11)
    np.testing.assert_allclose(sim.phs_var_subharm, g["phs_var_subharm"], rtol=1e-11)
    np.testing.assert_allclose(sim.phs_var_weights_sh, g["phs_var_weights_sh"], rtol=1e-11)
    np.testing.assert_allclose(np.asarray(sim.lf_mask_subharm, dtype=float), g["lf_mask_subharm"], rtol=1e-12, atol=1e-15)
    q = params_from_json(load_golden("e2e_ao_alias")["params_json"])
    q.update({"GPU_DEVICE": 0})
    plain = fast_amd.Fast(q)
    assert plain.powerspec_subharm is None and plain.phs_var_subharm is None and plain.phs_var_weights_sh is None


# ------------------------------------------------------------------ chirp-z family: grid sizes that are not 64 P
@pytest.mark.parametrize("N,Np", [(164, 82), (102, 40), (49, 23), (33, 9), (252, 129), (502, 82), (943, 82), (1002, 82), (1455, 82),
                                  (1280 - 255, 256), (1900, 100), (333, 333 // 3),
                                  (2200, 82), (2816, 82), (2050, 200), (4090, 82), (1971, 129)])      # rows in input blocks (N + Np - 1 > 2048)
@pytest.mark.parametrize("prec,tol", [("f64", 1e-11), ("f32", 5e-5)])
def test_chirpz_kernels_match_oracle_fft(N, Np, prec, tol):
    """Arbitrary N (the reference auto-sizes to e.g. 164, fast.py:176-211; odd N with numpy's asymmetric fftshift) on the
    chirp-z kernels: screens from host coefficients against the oracle's FFT-branch transform, and against the direct family."""
    if N > 2048 and prec == "f32":
        pytest.skip("one precision is enough for the blocked rows of the largest grids")
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    rng = np.random.default_rng(N + Np)
    nb = 2 if N <= 2048 else 1
    cr, ci = rng.normal(size=(nb, N, N)), rng.normal(size=(nb, N, N))
    want = R.crop(R.double_screens(R.screens_fftw((cr + 1j * ci) * np.sqrt(ps), df)), N, Np)
    for lo in ((N - Np) // 2, 0, N - Np):
        h = _lib.Handle(N, Np, prec, 0)
        assert h.kernel_path() == (2 if N >= 96 else 0)
        h.kernel_path(2)
        h.set_spectrum(ps, df)
        h.set_pupil(np.ones((Np, Np)), lo, 0.01)
        got = h.screens_coeffs(cr, ci)
        if lo != (N - Np) // 2:
            full = R.double_screens(R.screens_fftw((cr + 1j * ci) * np.sqrt(ps), df))
            want_lo = full[:, lo:lo + Np, lo:lo + Np]
        else:
            want_lo = want
        assert np.abs(got - want_lo).max() <= tol * np.abs(want).max()
    h.kernel_path(0)
    got_d = h.screens_coeffs(cr[:1], ci[:1])
    assert np.abs(got_d[0] - want_lo[0]).max() <= tol * np.abs(want).max()


@pytest.mark.parametrize("N,Np", [(164, 82), (1002, 82), (302, 150), (2200, 82), (2816, 140)])
def test_chirpz_device_generator_equals_direct_family(N, Np):
    h, ps, df, W = _small_problem(N, Np)
    assert h.kernel_path() == 2
    a = h.run(7, 3, 6, None, 0.02)
    coh = h.run(7, 3, 6, None, 0.02, coherent=True)
    np.testing.assert_allclose(np.abs(coh) ** 2, a, rtol=1e-12)
    h.set_batch(4)
    np.testing.assert_array_equal(h.run(7, 3, 6, None, 0.02), a)
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(7, 3, 6, None, 0.02), a, rtol=1e-9)
    # and the restated generator + oracle
    coeffs = np.stack([devrng.device_coefficients(7, 3 + j, N) for j in range(6)])
    chi = devrng.device_logamp_normals(7, 6, 12) * np.sqrt(0.02)
    la = np.concatenate([chi[0::2], chi[1::2]])
    np.testing.assert_allclose(a, R.powers_from_coefficients(coeffs, ps, df, W, 0.01, la), rtol=DEVICE_RTOL)


@pytest.mark.parametrize("case", ["default164", "oddN", "oddNp", "autosize", "subharm"])
def test_fast_run_reproduces_reference_on_chirpz_kernels(case):
    """The reference's own grids (auto-sized 164, odd 49, 48, 64) forced onto the chirp-z family: same SEED -> same `_r`."""
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "host", "GPU_DEVICE": 0, "GPU_KERNELS": "chirpz"})
    sim = fast_amd.Fast(p)
    assert sim._handle.kernel_path() == 2
    np.testing.assert_allclose(sim.run()._r, g["r"], rtol=1e-9)


# ------------------------------------------------------------------ 50-lane family: N = 50 P (100, 200, 250, 500, 1000, ...)
@pytest.mark.parametrize("N,Np", [(100, 40), (150, 120), (200, 128), (250, 82), (300, 33), (350, 82), (400, 256), (450, 82), (500, 82),
                                  (600, 200), (700, 82), (800, 101), (900, 82), (1000, 82), (1000, 256), (1200, 82), (1400, 82),
                                  (1600, 128), (1350, 82), (1500, 82), (1750, 60), (2000, 82), (2000, 200), (2500, 82), (3000, 101),
                                  (4000, 82)])
@pytest.mark.parametrize("prec,tol", [("f64", 1e-11), ("f32", 5e-5)])
def test_lanes50_kernels_match_oracle_fft(N, Np, prec, tol):
    """Round decimal grids (NPXLS 1000 etc.) on the 50-lane mixed-radix kernels (fmc_mrfft.h): screens from host coefficients
    against the oracle's FFT-branch transform for windows in the middle and at both ends, and against the direct family."""
    if N > 2000 and prec == "f32":
        pytest.skip("one precision is enough for the largest split grids")
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    rng = np.random.default_rng(N + Np)
    nb = 2 if N <= 2000 else 1
    cr, ci = rng.normal(size=(nb, N, N)), rng.normal(size=(nb, N, N))
    full = R.double_screens(R.screens_fftw((cr + 1j * ci) * np.sqrt(ps), df))
    for lo in sorted({(N - Np) // 2, 0, N - Np}):
        h = _lib.Handle(N, Np, prec, 0)
        assert h.kernel_path() == 3
        h.set_spectrum(ps, df)
        h.set_pupil(np.ones((Np, Np)), lo, 0.01)
        got = h.screens_coeffs(cr, ci)
        want = full[:, lo:lo + Np, lo:lo + Np]
        assert np.abs(got - want).max() <= tol * np.abs(full).max()
    h.kernel_path(0)
    got_d = h.screens_coeffs(cr[:1], ci[:1])
    assert np.abs(got_d[0] - want[0]).max() <= tol * np.abs(full).max()
    with pytest.raises(_lib.FastMCError):
        h.kernel_path(2)             # 50 streams per row: the chirp-z kernels (64 streams) do not serve these grids


@pytest.mark.parametrize("N,Np", [(100, 50), (500, 82), (1000, 82), (600, 150), (1400, 82), (2500, 82)])
def test_lanes50_device_generator_equals_direct_family(N, Np):
    h, ps, df, W = _small_problem(N, Np)
    assert h.kernel_path() == 3
    a = h.run(7, 3, 6, None, 0.02)
    coh = h.run(7, 3, 6, None, 0.02, coherent=True)
    np.testing.assert_allclose(np.abs(coh) ** 2, a, rtol=1e-12)
    h.set_batch(4)
    np.testing.assert_array_equal(h.run(7, 3, 6, None, 0.02), a)
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(7, 3, 6, None, 0.02), a, rtol=1e-9)
    # and the restated generator (50 streams per row) + oracle
    coeffs = np.stack([devrng.device_coefficients(7, 3 + j, N) for j in range(6)])
    chi = devrng.device_logamp_normals(7, 6, 12) * np.sqrt(0.02)
    la = np.concatenate([chi[0::2], chi[1::2]])
    np.testing.assert_allclose(a, R.powers_from_coefficients(coeffs, ps, df, W, 0.01, la), rtol=DEVICE_RTOL)


@pytest.mark.parametrize("case", ["npxls100", "npxls150", "npxls200"])
def test_fast_run_reproduces_reference_on_lanes50_and_direct_kernels(case):
    """The captured 100^2 / 150^2 / 200^2 runs of the reference: same SEED -> same `_r` on the 50-lane family (the default there)
    and on the direct family."""
    g = load_golden("e2e_" + case)
    for fam, path in (("auto", 3), ("direct", 0)):
        p = params_from_json(g["params_json"])
        p.update({"GPU_RNG": "host", "GPU_DEVICE": 0, "GPU_KERNELS": fam})
        sim = fast_amd.Fast(p)
        assert sim._handle.kernel_path() == path
        np.testing.assert_allclose(sim.run()._r, g["r"], rtol=1e-9)


@pytest.mark.parametrize("N,Np", [(1344, 82), (1920, 200), (2304, 82), (2560, 101), (3072, 82), (3072, 256), (3584, 60), (3840, 82)])
def test_wave_family_with_run_time_sub_rows(N, Np):
    """Grids N = 64 P S whose sub-row count is not a compiled one (fmc_core.h: wave_rt_split): screens from host coefficients
    vs the oracle's FFT-branch transform, device-generator powers vs the direct family (64 S streams per row), and the
    restated generator."""
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    rng = np.random.default_rng(N + Np)
    cr, ci = rng.normal(size=(1, N, N)), rng.normal(size=(1, N, N))
    full = R.double_screens(R.screens_fftw((cr + 1j * ci) * np.sqrt(ps), df))
    for lo in sorted({(N - Np) // 2, 0, N - Np}):
        h = _lib.Handle(N, Np, "f64", 0)
        assert h.kernel_path() == 1
        h.set_spectrum(ps * 0.02, df)
        h.set_pupil(_window_W(Np), lo, 0.01)
        got = h.screens_coeffs(cr, ci)
        assert np.abs(got - full[:, lo:lo + Np, lo:lo + Np] * np.sqrt(0.02)).max() <= 1e-11 * np.abs(full).max()
    a = h.run(7, 3, 2, None, 0.02)
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(7, 3, 2, None, 0.02), a, rtol=1e-9)
    with pytest.raises(_lib.FastMCError):
        h.kernel_path(2)
    got = h.rng_coeffs(7, 3)
    assert np.abs(got - devrng.device_coefficients(7, 3, N)).max() < 1e-3


@pytest.mark.parametrize("N,Np,lo", [(1024, 40, None), (1024, 82, None), (1024, 96, None), (1024, 97, None), (1024, 128, None),
                                     (1024, 200, None), (1024, 256, None), (1024, 400, None), (1024, 82, 0), (1024, 82, 500),
                                     (1024, 120, 904), (2048, 82, None), (2048, 122, None), (2048, 402, None), (4096, 82, None)])
def test_p16_row_variants_equal_the_direct_family(N, Np, lo):
    """P = 16 grids pick their row by window: the 16 x 4 lane factorisation with six of sixteen planes (centred, <= 96 pixels; dense
    at 1024^2), eight planes (97-128), all planes (anything else, NS = 2 / 4 / 8).  Each against the direct family on the same
    device draws, and the screens of host coefficients against numpy."""
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    lo = (N - Np) // 2 if lo is None else lo
    h = _lib.Handle(N, Np, "f64", 0)
    h.set_spectrum(ps * 0.02, df)
    h.set_pupil(_window_W(Np), lo, 0.01)
    assert h.kernel_path() == 1
    n = 2 if N <= 2048 else 1
    a = h.run(11, 5, n, None, 0.02)
    h.set_batch(1)
    np.testing.assert_array_equal(h.run(11, 5, n, None, 0.02), a)
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(11, 5, n, None, 0.02), a, rtol=1e-9)
    if N == 1024:
        h.kernel_path(1)
        rng = np.random.default_rng(Np)
        cr, ci = rng.normal(size=(1, N, N)), rng.normal(size=(1, N, N))
        z = np.fft.fftshift(np.fft.fft2(np.fft.fftshift((cr[0] + 1j * ci[0]) * np.sqrt(ps * 0.02) * df)))[lo:lo + Np, lo:lo + Np]
        got = h.screens_coeffs(cr, ci)
        assert max(np.abs(got[0] - z.real).max(), np.abs(got[1] - z.imag).max()) <= 1e-11 * np.abs(z).max()


# Every device-generator instantiation a dispatch can reach, each DIRECTLY against the oracle (not through another HIP
# kernel): fastmc.hip:dispatch_wave picks the row by (P, window): at P = 16 the 16 x 4 lane factorisation with six planes
# (centred window <= 96 pixels; dense sixteen-wave kernels at 1024^2 -- the BENCHMARKED instantiation
# k_rows_wave<double,16,2,0,1,4> --, twelve-wave / split rows at 2048^2 and 4096^2), eight planes (97-128 pixels), all
# sixteen planes (any other window: NS = 2, 4, 8), the dense 8 x 8 row with all eight / six of eight planes for off-centre
# windows; P = 18, 20, 24, 28 without the planes a centred window never reads; the 50-lane and run-time-split rows with and
# without pruned planes; chirp-z and the general rows of the other sizes.
_VARIANTS = [(1024, 40, None), (1024, 82, None), (1024, 96, None), (1024, 97, None), (1024, 128, None), (1024, 200, None),
             (1024, 256, None), (1024, 400, None), (1024, 82, 0), (1024, 82, 340), (1024, 82, 500), (1024, 120, 904),
             (2048, 82, None), (2048, 122, None), (2048, 402, None), (4096, 82, None),
             (1152, 82, None), (1280, 82, None), (1536, 82, None), (1792, 82, None), (512, 82, None), (256, 82, None), (768, 152, None),
             (1000, 82, None), (2000, 82, None), (1200, 82, None), (500, 82, None), (3072, 82, None), (1344, 82, None),
             (164, 82, None), (943, 82, None),
             # packed rows (eight / four / two rows per wavefront): six centred planes, all planes, off-centre and wide windows, the whole
             # grid, and a window beyond the packed kernels (512, 300: device draws go to the direct family)
             (128, 82, None), (128, 96, None), (128, 97, None), (128, 40, 0), (128, 128, None), (128, 60, 68),
             (256, 96, None), (256, 97, None), (256, 40, 0), (256, 82, 100), (256, 200, None), (256, 256, None),
             (512, 96, None), (512, 128, None), (512, 82, 3), (512, 250, 7), (512, 256, 256), (512, 300, None)]


@pytest.mark.parametrize("N,Np,lo", _VARIANTS)
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_every_device_mode_row_variant_matches_the_oracle(N, Np, lo, prec):
    if prec == "f32" and (N > 2048 or (N, Np, lo) not in [(1024, 82, None), (1024, 128, None), (1024, 200, None), (2048, 82, None), (1000, 82, None), (512, 82, None),
                                                         (256, 82, None), (256, 200, None), (512, 250, 7), (128, 82, None), (128, 128, None)]):
        pytest.skip("float32 pipeline: the benchmarked shapes only")
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    ps = ps * 0.02
    lo = (N - Np) // 2 if lo is None else lo
    W = _window_W(Np)
    h = _lib.Handle(N, Np, prec, 0)
    h.set_spectrum(ps, df)
    h.set_pupil(W, lo, 0.01)
    seed, real0, n = 2026, 7, (2 if N <= 1536 else 1)
    got = h.run(seed, real0, n, None, 0.01)
    # (1) against the RESTATEMENT of the generator: nothing in `want` comes from the device, so a variant whose fused draw and
    #     read-back agreed with each other and both differed from the definition would fail here
    want = _oracle_powers_from_restated_draws(seed, real0, n, ps, df, W, lo, 0.01, 0.01)
    assert (want > 1e-3).all()                                  # few-radian screens: no deep fade amplifies the rounding
    np.testing.assert_allclose(got, want, rtol=DEVICE_RTOL if prec == "f64" else 2e-4)
    # (2) against the oracle on the device's own read-back draws (fastmc_rng_coeffs, float32 colouring as the kernels colour):
    #     tighter, because the hardware transcendentals' ~1e-7 cancels
    want_rb = _oracle_powers_from_device_draws(h, seed, real0, n, ps, df, W, lo, 0.01, 0.01)
    np.testing.assert_allclose(got, want_rb, rtol=DEVICE_RTOL if prec == "f64" else 2e-4)
    if prec == "f64":
        assert np.abs(got / want_rb - 1).max() < 2e-6           # what is actually observed: ~1e-7


def test_benchmarked_instantiation_at_baseline_size_matches_the_oracle():
    """BASELINE configs[1] as bench.py runs it (1024^2, Np = 82, NOAO von Karman spectrum of the HV5/7 profile at 55 deg,
    L0 = inf: 13 rad rms screens, deep fades included): 16 iterations of k_rows_wave<double,16,2,0,1,4> + its column kernel
    against the oracle on the device's own draws.  Bar 1e-5 relative to the MEAN power (a fade of 1e-4 of the mean amplifies
    any rounding 1e4-fold in relative terms) and 1e-4 on every single power."""
    g = load_golden("big_noao_1024")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "GPU_RNG": "device", "NITER": 16, "NCHUNKS": 1, "SEED": 77})
    sim = fast_amd.Fast(p)
    assert sim.Npxls == 1024 and sim.Npxls_pup == 82 and sim._handle.kernel_path() == 1
    r = sim.run()._r
    h = sim._handle
    lo = int(sim._prob.pup.crop_lo)
    want = _oracle_powers_from_device_draws(h, 77, 0, 8, sim.powerspec, sim._prob.df, sim._prob.W, lo, sim.dx, float(sim.logamp_var))
    assert np.abs(r - want).max() < 1e-5 * want.mean()
    np.testing.assert_allclose(r, want, rtol=1e-4)


@pytest.mark.parametrize("N,Np,sub", [(512, 82, False), (1024, 82, False), (1000, 82, False), (2048, 82, False), (164, 82, False), (256, 40, True),
                                      (128, 40, True), (512, 82, True), (256, 200, True), (128, 128, False)])
def test_float64_device_generator_matches_its_restatement(N, Np, sub):
    """GPU_RNG_PRECISION 'f64' (fastmc_set_rng_precision): 53-bit normals, float64 log / sqrt / sincospi, float64 colouring
    -- the reference's precision (funcs.py:352-356, fast.py:594).  Its draws equal oracle/devrng.py's restatement to a few
    ulp, the powers equal the oracle's on those draws to the float64 pipeline's bar (1e-9), on every kernel family, whatever
    the batch; and the float32 generator's powers for the same seed differ from it by what the float32 shortcut costs."""
    h, ps, df, W = _small_problem(N, Np)
    h.set_rng_precision("f64")
    seed, real0, n = 31, 4, (3 if N <= 1024 else 1)
    for g in (real0, 2 ** 33 + 1):
        assert np.abs(h.rng_coeffs(seed, g) - devrng.device_coefficients_f64(seed, g, N)).max() < 2e-14
    la_n = h.rng_logamp(seed, 2 * real0, 2 * n)
    assert np.abs(la_n - devrng.device_logamp_normals(seed, 2 * real0, 2 * n, f64=True)).max() < 2e-14
    sub_args = None
    if sub:
        grid = R.subharm_grid(N, 0.01)
        ps_lo = np.random.default_rng(1).uniform(0.5, 2.0, size=(3, 3, 3)) * 1e-3
        h.set_subharm(ps_lo, grid.fx, grid.fy, grid.df)
        rand_lo = np.stack([devrng.device_subharm_coefficients(seed, real0 + j, f64=True) for j in range(n)])
        sub_args = (rand_lo, ps_lo, grid)
    got = h.run(seed, real0, n, None, 0.01)
    coeffs = np.stack([devrng.device_coefficients_f64(seed, real0 + j, N) for j in range(n)])
    chi = devrng.device_logamp_normals(seed, 2 * real0, 2 * n, f64=True) * 0.1
    la = np.concatenate([chi[0::2], chi[1::2]])
    want = R.powers_from_coefficients(coeffs, ps, df, W, 0.01, la, sub=sub_args)
    np.testing.assert_allclose(got, want, rtol=1e-9)
    h.set_batch(2)
    np.testing.assert_array_equal(h.run(seed, real0, n, None, 0.01), got)
    h.set_batch(0)
    h.run_async(seed, real0, n, 0.01)
    np.testing.assert_array_equal(h.wait(), got)
    fam = h.kernel_path()
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(seed, real0, n, None, 0.01), got, rtol=1e-9)
    h.kernel_path(fam)
    scr = h.screens(seed, real0, 1)
    z = R.screens_fftw(coeffs[:1] * np.sqrt(ps), df)
    lo = (N - Np) // 2
    if not sub:
        assert np.abs(scr[0] - z[0].real[lo:lo + Np, lo:lo + Np]).max() < 1e-10 * np.abs(z).max()
    # the float32 generator on the same seed: the same normals to ~2^-24
    h.set_rng_precision("f32")
    f32 = h.run(seed, real0, n, None, 0.01)
    assert np.abs(f32 / got - 1).max() < DEVICE_RTOL
    err = np.abs(h.rng_coeffs(seed, real0) - coeffs[0])          # 24-bit u: the radius loses relative accuracy where u -> 1
    assert err.max() < 1e-3 and np.quantile(err, 0.9999) < 1e-5


# every MODE 2 instantiation a dispatch reaches (fastmc.hip: dispatch_wave): dense six / eight planes, twelve-wave six / eight /
# sixteen planes, NS = 4 / 8, split rows of 2048 / 4096, off-centre windows; the packed rows; every other P of the family
_FUSED64 = [(1024, 40, None, "k_rows_wave<double, 16, 2, 2, 1, 4>"), (1024, 96, None, "k_rows_wave<double, 16, 2, 2, 1, 4>"),
            (1024, 97, None, "k_rows_wave<double, 16, 2, 2, 1, 8>"), (1024, 128, None, "k_rows_wave<double, 16, 2, 2, 1, 8>"),
            (1024, 82, 0, "k_rows_wave<double, 16, 2, 2, 1, 7>"), (1024, 120, 904, "k_rows_wave<double, 16, 2, 2, 1, 7>"),
            (1024, 200, None, "k_rows_wave<double, 16, 4, 2, 1, 7>"), (1024, 400, None, "k_rows_wave<double, 16, 8, 2, 1, 7>"),
            (2048, 82, None, "k_rows_wave<double, 16, 2, 2, 2, 4>"), (2048, 122, None, "k_rows_wave<double, 16, 2, 2, 2, 6>"),
            (2048, 402, None, "k_rows_wave<double, 16, 8, 2, 2, 7>"), (4096, 82, None, "k_rows_wave<double, 16, 2, 2, 4, 4>"),
            # packed rows (eight / four / two rows per wavefront): centred six planes and all planes
            (128, 82, None, "k_rows_pk<double, 0, 2, 0>"), (128, 128, None, "k_rows_pk<double, 0, 2, 1>"),
            (256, 82, None, "k_rows_pk<double, 1, 2, 0>"), (256, 200, None, "k_rows_pk<double, 1, 2, 1>"), (256, 82, 100, "k_rows_pk<double, 1, 2, 1>"),
            (512, 96, None, "k_rows_pk<double, 2, 2, 0>"), (512, 250, 7, "k_rows_pk<double, 2, 2, 1>"),
            # the other one-row-per-wave grids (192 ... 1792): the plain variant, windows of up to 128 / 256 pixels
            (192, 30, None, "k_rows_wave<double, 3, 2, 2, 1, 0>"), (320, 82, 3, "k_rows_wave<double, 5, 2, 2, 1, 0>"),
            (384, 60, None, "k_rows_wave<double, 6, 2, 2, 1, 0>"), (448, 128, None, "k_rows_wave<double, 7, 2, 2, 1, 0>"),
            (576, 82, None, "k_rows_wave<double, 9, 2, 2, 1, 0>"), (576, 200, None, "k_rows_wave<double, 9, 4, 2, 1, 0>"),
            (640, 82, None, "k_rows_wave<double, 10, 2, 2, 1, 0>"), (640, 250, 11, "k_rows_wave<double, 10, 4, 2, 1, 0>"),
            (768, 82, None, "k_rows_wave<double, 12, 2, 2, 1, 0>"), (768, 256, None, "k_rows_wave<double, 12, 4, 2, 1, 0>"),
            (896, 100, None, "k_rows_wave<double, 14, 2, 2, 1, 0>"), (1152, 82, None, "k_rows_wave<double, 18, 2, 2, 1, 0>"),
            (1280, 82, None, "k_rows_wave<double, 20, 2, 2, 1, 0>"), (1280, 200, None, "k_rows_wave<double, 20, 4, 2, 1, 0>"),
            (1536, 120, 1400, "k_rows_wave<double, 24, 2, 2, 1, 0>"), (1536, 222, None, "k_rows_wave<double, 24, 4, 2, 1, 0>"),
            (1792, 82, None, "k_rows_wave<double, 28, 2, 2, 1, 0>"),
            # 50-lane family (N = 50 P S) and the run-time-split wave grids: the rows of fmc_mrfft.h
            (100, 40, None, "k_rows_mr<double, 2, 2, 2, false, 50, 0>"), (300, 60, None, "k_rows_mr<double, 6, 2, 2, false, 50, 0>"),
            (500, 82, None, "k_rows_mr<double, 10, 2, 2, false, 50, 0>"), (800, 96, 3, "k_rows_mr<double, 16, 2, 2, false, 50, 0>"),
            (1000, 82, None, "k_rows_mr<double, 20, 2, 2, false, 50, 0>"), (1000, 200, None, "k_rows_mr<double, 20, 4, 2, false, 50, 0>"),
            (1200, 100, None, "k_rows_mr<double, 24, 2, 2, false, 50, 0>"), (2000, 82, None, "k_rows_mr<double, 20, 2, 2, true, 50, 0>"),
            (1750, 70, None, "k_rows_mr<double, 7, 2, 2, true, 50, 0>"), (1344, 82, None, "k_rows_mr<double, 7, 2, 2, true, 64, 0>"),
            (2560, 120, None, "k_rows_mr<double, 20, 2, 2, true, 64, 0>"),
            # chirp-z family (any other N): one transform of length 64 P, and rows in input blocks beyond 2048
            (164, 60, None, "k_rows_blu<double, 4, 2, 2, false>"), (291, 82, 5, "k_rows_blu<double, 8, 2, 2, false>"),
            (722, 200, None, "k_rows_blu<double, 16, 4, 2, false>"), (1111, 82, None, "k_rows_blu<double, 24, 2, 2, false>"),
            (1901, 82, None, "k_rows_blu<double, 32, 2, 2, false>"), (3901, 82, None, "k_rows_blu<double, 16, 2, 2, true>")]


@pytest.mark.parametrize("N,Np,lo,kernel", _FUSED64)
def test_fused_float64_generator_rows_match_the_oracle_on_restated_draws(N, Np, lo, kernel):
    """MODE 2 of the row kernels of every FFT family (fmc_kernels.h): the float64 generator drawn inside the row.  Powers against the
    oracle on oracle/devrng.py's float64 restatement at the float64 pipeline's bar (1e-9: nothing float32 is left in the
    path), the kernel that ran is the fused one, and the staged form (FASTMC_GEN64_STAGED: k_gen_coeffs_f64 -> MODE 1 rows)
    is covered by test_float64_device_generator_matches_its_restatement on the other families."""
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    ps = ps * 0.02
    lo = (N - Np) // 2 if lo is None else lo
    W = _window_W(Np)
    h = _lib.Handle(N, Np, "f64", 0)
    h.set_spectrum(ps, df)
    h.set_pupil(W, lo, 0.01)
    h.set_rng_precision("f64")
    seed, real0, n = 99, 2 ** 33 + 6, (2 if N <= 1024 else 1)
    got = h.run(seed, real0, n, None, 0.01)
    assert h.last_kernels()[0] == kernel
    amp = np.sqrt(ps) * df
    re, im = [], []
    for j in range(n):
        z = R.screens_fftw(devrng.device_coefficients_f64(seed, real0 + j, N) * amp, 1.0)[lo:lo + Np, lo:lo + Np]
        re.append(z.real)
        im.append(z.imag)
    chi = devrng.device_logamp_normals(seed, 2 * real0, 2 * n, f64=True) * 0.1
    want = R.detector(np.stack(re + im), W, 0.01, np.concatenate([chi[0::2], chi[1::2]]))
    assert (want > 1e-3).all()
    np.testing.assert_allclose(got, want, rtol=1e-9)
    # the read-back is the fused rows' arithmetic bit for bit: direct family (staged draws) on the same seed agrees to rounding
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(seed, real0, n, None, 0.01), got, rtol=1e-10)


def test_fast_object_with_the_float64_generator():
    """The DEFAULT of `Fast(config).run()` on a float64 handle is the generator at the reference's precision (GPU_RNG_PRECISION
    'auto' = 'f64'); the opt-in float32 draw of the same seed has the same distribution (the same normals to
    2^-24: the vectors agree to ~1e-6 here); sharded over two handles identical to one; a float32 pipeline draws in float32."""
    g = load_golden("e2e_ao_alias")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "NITER": 400, "NCHUNKS": 4, "SEED": 5, "GPU_RNG": "device"})
    s32 = fast_amd.Fast(dict(p, GPU_RNG_PRECISION="f32"))
    r32 = s32.run()._r
    sim = fast_amd.Fast(dict(p))
    assert sim.rng_precision == "f64" and s32.rng_precision == "f32"
    r64 = sim.run()._r
    assert np.array_equal(fast_amd.Fast(dict(p, GPU_RNG_PRECISION="f64")).run()._r, r64)
    # 'auto' follows the precision the handle COMPUTES in: a float32 pipeline draws in float32 where the grid has float32 kernels
    # (NPXLS 256); a grid without them is promoted to float64 (fastmc_create) and then draws in float64 too
    f32 = fast_amd.Fast(dict(p, GPU_PRECISION="f32", NPXLS=256))
    assert f32.precision == "f32" and f32.rng_precision == "f32"
    promoted = fast_amd.Fast(dict(p, GPU_PRECISION="f32"))
    assert promoted.rng_precision == ("f32" if promoted.precision == "f32" else "f64")
    assert np.isfinite(r64).all() and not np.array_equal(r32, r64)
    np.testing.assert_allclose(r32, r64, rtol=1e-4)
    p2 = dict(p, GPU_RNG_PRECISION="f64", GPU_DEVICES=[0, 0])
    p2.pop("GPU_DEVICE")
    assert np.array_equal(fast_amd.Fast(p2).run()._r, r64)
    # the log-amplitudes on the object are the float64 draws
    chi = devrng.device_logamp_normals(5, 0, 400, f64=True) * np.sqrt(sim.logamp_var)
    half = 50
    la = np.empty((4, 100))
    la[:, :half], la[:, half:] = chi[0::2].reshape(4, half), chi[1::2].reshape(4, half)
    np.testing.assert_allclose(sim.logamp, la.ravel(), rtol=1e-12, atol=1e-15)
    with pytest.raises(Exception, match="GPU_RNG_PRECISION"):
        fast_amd.Fast(dict(p, GPU_RNG_PRECISION="f16"))


def test_float32_generator_shortcut_is_bounded_on_identical_draws():
    """What the float32 draws + float32 colouring of device mode cost in accuracy, isolated from the generator: the SAME numpy
    draws through the float64 pipeline once as float64 coefficients (the reference's arithmetic, fast.py:594) and once rounded
    to float32 and coloured in float32 (what fmc_kernels.h:draw_coloured does with its own draws), at BASELINE configs[1]
    (1024^2, NOAO, L0 = 25 m: 26 rad rms).  Recorded in DESIGN.md section 2."""
    g = load_golden("big_noao_L0_1024")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0})
    sim = fast_amd.Fast(p)
    N, h = sim.Npxls, sim._handle
    ps, df = sim.powerspec, sim._prob.df
    rng = np.random.default_rng(5)
    B = 8
    cr, ci = rng.normal(size=(B, N, N)), rng.normal(size=(B, N, N))
    la = np.zeros(2 * B)
    p64 = h.run_coeffs(cr, ci, la)
    # float32 draws, float32 colouring, widened: fed as "coefficients" of a unit spectrum so that nothing else multiplies them
    amp32 = (np.sqrt(ps) * df).astype(np.float32)
    c32r = (cr.astype(np.float32) * amp32).astype(np.float64) / df
    c32i = (ci.astype(np.float32) * amp32).astype(np.float64) / df
    h.set_spectrum(np.ones((N, N)), df)
    p32 = h.run_coeffs(c32r, c32i, la)
    rel = np.abs(p32 / p64 - 1)
    scr = sim._handle.screens_coeffs(c32r[:1], c32i[:1])
    assert np.abs(scr).max() > 10.0                              # tens of radians: the hard case
    print(f"float32 draw + colouring vs float64 on identical draws: max rel {rel.max():.2e}, median {np.median(rel):.2e}, "
          f"max abs / mean power {np.abs(p32 - p64).max() / p64.mean():.2e}")
    assert np.abs(p32 - p64).max() < 2e-5 * p64.mean() and np.median(rel) < 2e-5


def test_kernel_family_notes_in_the_log(caplog):
    """Grids of the direct family are announced with the nearest fast sizes; 64 P and 50 P S grids are not."""
    import logging
    g = load_golden("e2e_npxls200")
    for npx, expect in ((200, False), (128, False), (202, False), (130, False)):
        p = params_from_json(g["params_json"])
        p.update({"GPU_DEVICE": 0, "NPXLS": npx, "NITER": 4, "NCHUNKS": 2})
        caplog.clear()
        with caplog.at_level(logging.WARNING):
            sim = fast_amd.Fast(p)
        assert any("direct O(N^2 Np)" in r.getMessage() for r in caplog.records) == expect
        assert sim._handle.kernel_path() == {200: 3, 128: 1, 202: 2, 130: 2}[npx]
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "NPXLS": 2200, "NITER": 2, "NCHUNKS": 1})            # 2200 = 50 x 44: chirp-z rows in three input blocks
    caplog.clear()
    with caplog.at_level(logging.WARNING):
        sim = fast_amd.Fast(p)
    assert sim._handle.kernel_path() == 2 and not any("direct O(N^2 Np)" in r.getMessage() for r in caplog.records)
    assert np.isfinite(sim.run()._r).all()
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "NPXLS": 1002, "D_GROUND": 2.6, "NITER": 2, "NCHUNKS": 1})   # a 262-pixel window outside the wave family
    caplog.clear()
    with caplog.at_level(logging.WARNING):
        sim = fast_amd.Fast(p)
    assert sim._handle.kernel_path() == 0 and any("direct O(N^2 Np)" in r.getMessage() for r in caplog.records)


def test_wide_windows_on_split_grids_use_the_wave_family():
    """Windows of 257-512 pixels (a 2.6-5 m aperture at 1 cm) at 2048^2 stay on the wave kernels (eight output slots per
    lane) with the device generator, and agree with the direct family."""
    h, ps, df, W = _small_problem(2048, 402)
    assert h.kernel_path() == 1
    a = h.run(5, 0, 2, None, 0.01)
    assert h.last_timing()["rows_launches"] == 1 and np.isfinite(a).all()
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(5, 0, 2, None, 0.01), a, rtol=1e-9)


def test_powerspec_attribute_can_be_replaced_like_in_the_reference():
    """`sim.powerspec` is fetched from the device on first use and, as in the reference (where run() reads the attribute
    in every chunk, fast.py:593-594), can be replaced before run(): the new grid colours the draws."""
    g = load_golden("e2e_ao_alias")
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "host", "GPU_DEVICE": 0})
    sim = fast_amd.Fast(dict(p))
    np.testing.assert_allclose(sim.powerspec, g["powerspec"], rtol=1e-10, atol=1e-13 * np.abs(g["powerspec"]).max())
    sim.powerspec = 4.0 * g["powerspec"]
    assert np.array_equal(sim.powerspec, 4.0 * g["powerspec"])
    r4 = sim.run()._r
    want = R.monte_carlo(p["SEED"], p["NITER"], p["NCHUNKS"], 4.0 * g["powerspec"], sim._prob.df, sim._prob.W, sim.dx, float(sim.logamp_var))
    np.testing.assert_allclose(r4, want, rtol=1e-9)
    with pytest.raises(ValueError):
        sim.powerspec = np.ones((3, 3))


def test_dense_sixteen_wave_kernels_equal_the_twelve_wave_kernels():
    """1024^2 with a window of up to 96 pixels runs the dense-image kernels (sixteen waves per workgroup); with
    FASTMC_NO_DENSE16=1 the same library keeps the twelve-wave kernels: same arithmetic, same results."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from tests.test_gpu_parity import _small_problem; "
            "h, ps, df, W = _small_problem(1024, 82); np.save(sys.argv[1], h.run(17, 2, 40, None, 0.02))") % ROOT
    outs = []
    for flag in ("0", "1"):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fastmc_dense_{flag}_{os.getpid()}.npy")
        env = dict(os.environ, FASTMC_NO_DENSE16=flag, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
        os.remove(path)
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-12)
    h, ps, df, W = _small_problem(1024, 82)
    np.testing.assert_array_equal(h.run(17, 2, 40, None, 0.02), outs[0 if not os.environ.get("FASTMC_NO_DENSE16") else 1])


# ------------------------------------------------------------------ grids beyond 4096 (the reference has no upper limit, fast.py:176-211)
@pytest.mark.parametrize("N,Np,kernel", [(4608, 60, "k_rows_mr<double, 24, 2, 0, true, 64, 2>"), (5000, 82, "k_rows_mr<double, 20, 2, 0, true, 50, 1>"),
                                         (7168, 100, "k_rows_mr<double, 16, 2, 0, true, 64, 0>"), (8192, 82, "k_rows_mr<double, 16, 2, 0, true, 64, 2>"),
                                         (4100, 82, "k_rows_blu<double, 16, 2, 0, true>"), (7003, 200, "k_rows_blu<double, 16, 4, 0, true>")])
def test_grids_beyond_4096_run_as_up_to_eight_sub_rows(N, Np, kernel):
    """N = 64 P S / 50 P S with a run-time sub-row count S <= 8 (fmc_core.h: wave_rt_split / mr_split) up to 8192, and any other
    N <= 8192 on the chirp-z kernels with its rows in up to eleven input blocks: screens from host coefficients against numpy's
    FFT; the device generator through the family's rows against the oracle on the restated draws (one size: the restatement is
    Python) and against the direct family on the same seed; the float64 generator fused in the rows against the direct family's
    staged draws."""
    rng = np.random.default_rng(N)
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    ps = ps * 0.02
    lo = (N - Np) // 2
    W = _window_W(Np)
    h = _lib.Handle(N, Np, "f64", 0)
    h.set_spectrum(ps, df)
    h.set_pupil(W, lo, 0.01)
    cr, ci = rng.normal(size=(1, N, N)), rng.normal(size=(1, N, N))
    a = h.screens_coeffs(cr, ci)
    z = np.fft.fftshift(np.fft.fft2(np.fft.fftshift((cr[0] + 1j * ci[0]) * np.sqrt(ps) * df)))[lo:lo + Np, lo:lo + Np]
    assert max(np.abs(a[0] - z.real).max(), np.abs(a[1] - z.imag).max()) < 1e-12 * np.abs(z).max()
    del cr, ci, z
    seed, real0 = 5, 2 ** 32 + 1
    got = h.run(seed, real0, 1, None, 0.01)
    assert h.last_kernels()[0] == kernel
    if N == 4608:
        want = _oracle_powers_from_restated_draws(seed, real0, 1, ps, df, W, lo, 0.01, 0.01)
        np.testing.assert_allclose(got, want, rtol=1e-5)
    h.set_rng_precision("f64")
    got64 = h.run(seed, real0, 1, None, 0.01)
    assert ", 2, true" in h.last_kernels()[0]                   # MODE 2: the float64 generator inside the row
    h.kernel_path(0)
    np.testing.assert_allclose(got64, h.run(seed, real0, 1, None, 0.01), rtol=1e-9)
    h.set_rng_precision("f32")
    np.testing.assert_allclose(got, h.run(seed, real0, 1, None, 0.01), rtol=1e-9)
    h.close()
