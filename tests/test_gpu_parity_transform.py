"""GPU parity, transforms: the screens of every kernel family (wave, packed, 50-lane, chirp-z, direct, run-time sub-rows) against the
reference's FFT known-answers and the oracle's transform; the free functions of fast_amd.funcs; sub-harmonic screens."""
from _parity import *      # noqa: F401,F403 (numpy, pytest, fixtures, fast_amd, the oracle, the shared helpers)

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ screens: golden FFT KATs (direct family)
@pytest.mark.parametrize("N", [16, 30, 33, 64, 128, 100, 150])
@pytest.mark.parametrize("prec,tol", [("f64", 1e-11), ("f32", 3e-5)])
def test_screens_match_reference_fft_kat(N, prec, tol):
    g = load_golden(f"kat_fft_N{N}")
    for Np in (N, max(1, N // 3), 5):
        lo = (N - Np) // 2
        h = f32_draw_handle(N, Np, prec, 0)
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
    h = f32_draw_handle(N, Np, prec, 0)
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
        h = f32_draw_handle(N, Np, "f64", 0)
        h.set_spectrum(ps, df)
        h.set_pupil(np.ones((Np, Np)), lo, 0.01)
        got = h.screens_coeffs(cr, ci)
        assert np.abs(got - full[:, lo:lo + Np, lo:lo + Np]).max() <= 1e-11 * np.abs(full).max()


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
        h = f32_draw_handle(N, Np, "f64", 0)
        h.kernel_path(path)
        h.set_spectrum(ps, df)
        h.set_pupil(np.ones((Np, Np)), lo, dx)
        h.set_subharm(ps_lo, grid.fx, grid.fy, grid.df)
        got = h.screens_coeffs(cr, ci, sr, si)
        assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()


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
        h = f32_draw_handle(N, Np, prec, 0)
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
        h = f32_draw_handle(N, Np, prec, 0)
        assert h.kernel_path() == 3
        h.set_spectrum(ps, df)
        h.set_pupil(np.ones((Np, Np)), lo, 0.01)
        got = h.screens_coeffs(cr, ci)
        want = full[:, lo:lo + Np, lo:lo + Np]
        assert np.abs(got - want).max() <= tol * np.abs(full).max()
    h.kernel_path(0)
    got_d = h.screens_coeffs(cr[:1], ci[:1])
    assert np.abs(got_d[0] - want[0]).max() <= tol * np.abs(full).max()
    # the chirp-z rows draw 64 streams per row, these grids 50 S: since round 6 the family can be forced for host / staged coefficients
    # (its device draws are staged -- fastmc.hip: family_streams_ok; tests/test_gpu_parity_generator.py has the draws)
    assert h.kernel_path(2) == 2
    got_b = h.screens_coeffs(cr[:1], ci[:1])
    assert np.abs(got_b[0] - want[0]).max() <= tol * np.abs(full).max()


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
        h = f32_draw_handle(N, Np, "f64", 0)
        assert h.kernel_path() == 1
        h.set_spectrum(ps * 0.02, df)
        h.set_pupil(_window_W(Np), lo, 0.01)
        got = h.screens_coeffs(cr, ci)
        assert np.abs(got - full[:, lo:lo + Np, lo:lo + Np] * np.sqrt(0.02)).max() <= 1e-11 * np.abs(full).max()
    a = h.run(7, 3, 2, None, 0.02)
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(7, 3, 2, None, 0.02), a, rtol=1e-9)
    # (the chirp-z family can be forced since round 6: host / staged coefficients only on a grid of another generator layout, so the
    # forced family still takes the same draws)
    assert h.kernel_path(2) == 2
    np.testing.assert_allclose(h.run(7, 3, 2, None, 0.02), a, rtol=1e-9)
    got = h.rng_coeffs(7, 3)
    assert np.abs(got - devrng.device_coefficients(7, 3, N)).max() < 1e-3


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
