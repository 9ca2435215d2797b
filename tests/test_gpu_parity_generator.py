"""GPU parity, device generator: the device's draws (both precisions) against oracle/devrng.py, every row variant a dispatch can
reach against the oracle on restated draws, family against family, the statistical battery.  (Split out of test_gpu_parity.py in round 5.)"""
from _parity import *      # noqa: F401,F403 (numpy, pytest, fixtures, fast_amd, the oracle, the shared helpers)

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ generator
@pytest.mark.parametrize("N", [16, 33, 128, 256, 512, 1024, 2048, 4096, 200, 1000, 1500, 2000])
def test_device_generator_matches_oracle_restatement(N):
    h = f32_draw_handle(N, max(1, N // 4), "f64", 0)
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


def test_device_generator_statistical_quality():
    """Moments, tails, uniformity of phase and independence across lanes / rows / realisations of the
    device generator (Philox-seeded xoshiro128+ streams + hardware Box-Muller), 4 x 1024^2 draws."""
    from scipy import stats
    h = f32_draw_handle(1024, 8, "f64", 0)
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


def _distribution_bars(r, ref, fade_dB):
    """The bars of VERDICT r5 item 3, the same at 256^2 and at the benchmarked 1024^2: two-sample KS p > 0.01 on the dB values,
    mean power within 3 standard errors, scintillation index within 10 %, fade probability (power below `fade_dB` relative to
    the reference's mean) within 3 sigma binomial."""
    from scipy import stats
    d1, d2 = 10 * np.log10(r), 10 * np.log10(ref)
    assert stats.ks_2samp(d1, d2).pvalue > 0.01
    se = np.sqrt(r.var() / r.size + ref.var() / ref.size)
    assert abs(r.mean() - ref.mean()) < 3 * se, (r.mean(), ref.mean(), se)
    si1, si2 = (r / r.mean()).var(), (ref / ref.mean()).var()
    assert abs(si1 / si2 - 1) < 0.10, (si1, si2)
    thr = ref.mean() * 10 ** (fade_dB / 10)
    p1, p2 = np.mean(r < thr), np.mean(ref < thr)
    pp = (np.sum(r < thr) + np.sum(ref < thr)) / (r.size + ref.size)
    assert pp * ref.size > 30, "the threshold must leave the reference a countable number of fades"
    assert abs(p1 - p2) < 3 * np.sqrt(pp * (1 - pp) * (1 / r.size + 1 / ref.size)), (p1, p2)


@pytest.mark.parametrize("key,pkey,fade_dB", [("r_ao", "params_json", -3.0), ("r_noao", "params2_json", -10.0)])
def test_device_mode_distribution_matches_reference_output(key, pkey, fade_dB):
    """20 000 GPU iterations with the DEFAULT device generator (float64 on a float64 handle) vs 10 000 iterations of the REFERENCE
    itself (numpy PCG64 draws; stat_ref_256 + stat_ref_256b, fast/fast.py:115-140, 949-983) on the same configuration at 256^2."""
    g, gb = load_golden("stat_ref_256"), load_golden("stat_ref_256b")
    ref = np.concatenate([g[key], gb[key]])
    p = params_from_json(g[pkey])
    p.update({"GPU_DEVICE": 0, "NITER": 20000, "NCHUNKS": 20, "SEED": 1234, "GPU_RNG": "device"})
    sim = fast_amd.Fast(p)
    r = sim.run()._r
    assert sim._handle.last_kernels()[0] == "k_rows_pk<double, 1, 2, 0>"        # MODE 2: the float64 generator drew in the row
    _distribution_bars(r, ref, fade_dB)


@pytest.mark.parametrize("key,pkey,fade_dB", [("r_noao", "params_json", -10.0), ("r_ao", "params2_json", -3.0)])
def test_default_generator_distribution_at_the_benchmarked_size(key, pkey, fade_dB):
    """the mode the bench line is quoted on (float64 device generator, the kernel k_rows_wave<double, 16, 2, 2, 1, 4>)
    at the size it is quoted on.  20 000 GPU iterations at 1024^2 vs 4000 iterations of the reference (stat_ref_1024: BASELINE
    configs[1] NOAO with L0 = 25 m -- 26 rad rms, deep fades -- and configs[2] AO + alias)."""
    g = load_golden("stat_ref_1024")
    ref = g[key]
    p = params_from_json(g[pkey])
    p.update({"GPU_DEVICE": 0, "NITER": 20000, "NCHUNKS": 20, "SEED": 4321, "GPU_RNG": "device"})
    sim = fast_amd.Fast(p)
    r = sim.run()._r
    assert sim._handle.last_kernels()[0] == "k_rows_wave<double, 16, 2, 2, 1, 4>"   # the headline kernel, MODE 2
    _distribution_bars(r, ref, fade_dB)


@pytest.mark.parametrize("N", [256, 1024, 2048])
def test_f32_pipeline_tracks_f64_on_the_same_device_draws(N):
    """Same seed -> same generator words in both precisions; only the transform arithmetic differs.
    Full-size check of the float32 pipeline against the float64 one (phases of tens of radians)."""
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    W = _window_W(82)
    out = {}
    for prec in ("f64", "f32"):
        h = f32_draw_handle(N, 82, prec, 0)
        h.set_spectrum(ps, df)
        h.set_pupil(W, (N - 82) // 2, 0.01)
        out[prec] = h.run(77, 3, 16, None, 0.01)
        scr = h.screens(77, 3, 1)
        out[prec + "_rms"] = scr.std()
    assert out["f64_rms"] > 0.5                      # radians rms over the window of one screen (piston-dominated: varies per draw)
    np.testing.assert_allclose(out["f32"], out["f64"], rtol=5e-4, atol=1e-9)


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


@pytest.mark.parametrize("N,Np,lo", [(1024, 40, None), (1024, 82, None), (1024, 96, None), (1024, 97, None), (1024, 128, None),
                                     (1024, 200, None), (1024, 256, None), (1024, 400, None), (1024, 82, 0), (1024, 82, 500),
                                     (1024, 120, 904), (2048, 82, None), (2048, 122, None), (2048, 402, None), (4096, 82, None)])
def test_p16_row_variants_equal_the_direct_family(N, Np, lo):
    """P = 16 grids pick their row by window: the 16 x 4 lane factorisation with six of sixteen planes (centred, <= 96 pixels; dense
    at 1024^2), eight planes (97-128), all planes (anything else, NS = 2 / 4 / 8).  Each against the direct family on the same
    device draws, and the screens of host coefficients against numpy."""
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    lo = (N - Np) // 2 if lo is None else lo
    h = f32_draw_handle(N, Np, "f64", 0)
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
             (768, 82, None), (768, 96, 333), (1280, 40, None), (1792, 82, 857), (1280, 82, 100),     # packed sub-rows; (768, 152) and (1280, 82, 100): beyond them
             (640, 82, None), (896, 82, None), (896, 96, 401), (1152, 64, None), (1536, 82, 730), (640, 82, 0),
             (576, 82, None), (576, 40, 250), (448, 82, None), (320, 96, None), (192, 82, None), (576, 82, 100), (384, 82, None),     # sub-rows of 64 points; the last: beyond them
             (1000, 82, None), (2000, 82, None), (1200, 82, None), (500, 82, None), (3072, 82, None), (1344, 82, None),
             (1920, 82, None), (2304, 96, None), (2560, 60, 1236), (1728, 82, None), (1920, 120, None),     # packed sub-rows with a run-time count (pks_rt); the last: beyond them (direct family)
             (704, 82, None), (1088, 40, 530), (1600, 82, None), (1664, 96, None), (832, 82, 5), (1600, 100, None),     # ... on chirp-z and 50-lane grids; the last two: beyond them (direct family)
             (164, 82, None), (943, 82, None), (650, 82, None), (1502, 60, None),      # chirp-z: M = 256, 1024, 768, 1792
             # packed rows (eight / four / two rows per wavefront): six centred planes, all planes, off-centre and wide windows, the whole
             # grid, and a window beyond the packed kernels (512, 300: device draws go to the direct family)
             (128, 82, None), (128, 96, None), (128, 97, None), (128, 40, 0), (128, 128, None), (128, 60, 68),
             (256, 96, None), (256, 97, None), (256, 40, 0), (256, 82, 100), (256, 200, None), (256, 256, None),
             (512, 96, None), (512, 128, None), (512, 82, 3), (512, 250, 7), (512, 256, 256), (512, 300, None)]


@pytest.mark.parametrize("N,Np,lo", _VARIANTS)
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_every_device_mode_row_variant_matches_the_oracle(N, Np, lo, prec):
    if prec == "f32" and (N > 2048 or (N, Np, lo) not in [(1024, 82, None), (1024, 128, None), (1024, 200, None), (2048, 82, None), (1000, 82, None), (512, 82, None),
                                                         (256, 82, None), (256, 200, None), (512, 250, 7), (128, 82, None), (128, 128, None), (768, 82, None), (1792, 82, None), (896, 82, None), (1536, 82, None), (576, 82, None), (320, 96, None)]):
        pytest.skip("float32 pipeline: the benchmarked shapes only")
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    ps = ps * 0.02
    lo = (N - Np) // 2 if lo is None else lo
    W = _window_W(Np)
    h = f32_draw_handle(N, Np, prec, 0)
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
                                      (128, 40, True), (512, 82, True), (256, 200, True), (128, 128, False),
                                      # packed sub-rows (round 6): rows AND columns in S passes (768, 896), rows only (576); sub-harmonics in their epilogues
                                      (768, 82, True), (896, 60, True), (576, 82, True), (1280, 82, False)])
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
            (2048, 122, None, "k_rows_pks<double, 1, -2, 2, 8>"), (2048, 122, 900, "k_rows_pks<double, 1, -2, 2, 16>"), (2048, 122, 600, "k_rows_wave<double, 16, 2, 2, 2, 7>"),
            (2048, 402, None, "k_rows_wave<double, 16, 8, 2, 2, 7>"), (4096, 100, None, "k_rows_pks<double, 1, -2, 2, 8>"),
            # ... whose centred windows of up to 96 pixels go to the packed sub-rows (eight / sixteen sub-rows of 256 points, count at run time:
            # the same N / 16 streams per row, +7 % / +15 %); 1024 keeps its dense sixteen-wave row (fastmc.hip: pks_p16_from)
            (2048, 82, None, "k_rows_pks<double, 1, -2, 2>"), (2048, 96, 976, "k_rows_pks<double, 1, -2, 2>"), (4096, 82, None, "k_rows_pks<double, 1, -2, 2>"),
            (2048, 82, 900, "k_rows_pks<double, 1, -2, 2, 16>"),
            # packed rows (eight / four / two rows per wavefront): centred six planes and all planes
            (128, 82, None, "k_rows_pk<double, 0, 2, 0>"), (128, 128, None, "k_rows_pk<double, 0, 2, 1>"),
            (256, 82, None, "k_rows_pk<double, 1, 2, 0>"), (256, 200, None, "k_rows_pk<double, 1, 2, 1>"), (256, 82, 100, "k_rows_pk<double, 1, 2, 1>"),
            (512, 96, None, "k_rows_pk<double, 2, 2, 0>"), (512, 250, 7, "k_rows_pk<double, 2, 2, 1>"),
            # the other one-row-per-wave grids (192 ... 1792): the plain variant, windows of up to 128 / 256 pixels
            # 192 / 320 / 448 / 576 (round 6): sub-rows of SIXTY-FOUR points, eight rows per wavefront (k_rows_pks<R, -1, S, MODE>); windows the 96
            # centred outputs do not hold are staged (MODE 1 rows)
            (192, 30, None, "k_rows_pks<double, -1, 3, 2>"), (192, 96, None, "k_rows_pks<double, -1, 3, 2>"), (320, 82, 3, "k_rows_wave<double, 5, 2, 1, 1, 0>"),
            (320, 82, None, "k_rows_pks<double, -1, 5, 2>"), (448, 60, 200, "k_rows_pks<double, -1, 7, 2>"),
            (384, 60, None, "k_rows_pks<double, 0, 3, 2>"), (384, 128, None, "k_rows_pks<double, 0, 0, 2, 8>"), (448, 128, None, "k_rows_pks<double, -1, 0, 2, 8>"),
            (576, 82, None, "k_rows_pks<double, -1, 9, 2>"), (576, 96, 240, "k_rows_pks<double, -1, 9, 2>"), (576, 200, None, "k_rows_wave<double, 9, 4, 1, 1, 0>"),
            (640, 82, None, "k_rows_pks<double, 0, 5, 2>"), (640, 96, 272, "k_rows_pks<double, 0, 5, 2>"), (640, 250, 11, "k_rows_wave<double, 10, 4, 1, 1, 0>"),
            # 640, 768, 896, 1152, 1280, 1536, 1792 (round 6): the packed sub-rows for centred windows of up to 96 pixels (also shifted inside the six planes);
            # any other window is STAGED (k_gen_coeffs_f64 -> MODE 1 rows: these grids draw N / 16 streams per row)
            (768, 82, None, "k_rows_pks<double, 1, 3, 2>"), (768, 40, None, "k_rows_pks<double, 1, 3, 2>"), (768, 256, None, "k_rows_pks<double, 1, 0, 2, 16>"),
            (768, 82, 0, "k_rows_wave<double, 12, 2, 1, 1, 0>"),
            (896, 100, None, "k_rows_pks<double, 0, 0, 2, 8>"), (896, 130, None, "k_rows_blu<double, 24, 4, 1, false>"), (896, 82, None, "k_rows_pks<double, 0, 7, 2>"), (1152, 82, None, "k_rows_pks<double, 0, 9, 2>"),
            (1152, 30, 545, "k_rows_pks<double, 0, 9, 2>"), (1152, 82, 500, "k_rows_wave<double, 18, 2, 1, 1, 0>"),
            (1280, 82, None, "k_rows_pks<double, 1, 5, 2>"), (1280, 96, None, "k_rows_pks<double, 1, 5, 2>"), (1280, 82, 602, "k_rows_pks<double, 1, 5, 2>"),
            (1280, 200, None, "k_rows_pks<double, 1, 0, 2, 16>"), (1280, 200, 100, "k_rows_wave<double, 20, 4, 1, 1, 0>"),
            (1536, 120, 1400, "k_rows_wave<double, 24, 2, 1, 1, 0>"), (1536, 222, None, "k_rows_pks<double, 1, -2, 2, 16>"), (1536, 82, None, "k_rows_pks<double, 1, 6, 2>"),
            (1536, 96, 720, "k_rows_pks<double, 1, 6, 2>"),
            (1792, 82, None, "k_rows_pks<double, 1, 7, 2>"), (1792, 60, 860, "k_rows_pks<double, 1, 7, 2>"), (1792, 97, None, "k_rows_pks<double, 1, 0, 2, 8>"), (1792, 97, 800, "k_rows_pks<double, 1, 0, 2, 16>"), (1792, 97, 300, "k_rows_wave<double, 28, 2, 1, 1, 0>"),
            # 50-lane family (N = 50 P S) and the run-time-split wave grids: the rows of fmc_mrfft.h
            (100, 40, None, "k_rows_mr<double, 2, 2, 2, false, 50, 0>"), (300, 60, None, "k_rows_mr<double, 6, 2, 2, false, 50, 0>"),
            (500, 82, None, "k_rows_mr<double, 10, 2, 2, false, 50, 0>"), (800, 96, 3, "k_rows_mr<double, 16, 2, 2, false, 50, 0>"),
            (1000, 82, None, "k_rows_mr<double, 20, 2, 2, false, 50, 0>"), (1000, 200, None, "k_rows_mr<double, 20, 4, 2, false, 50, 0>"),
            (1200, 100, None, "k_rows_mr<double, 24, 2, 2, false, 50, 0>"), (2000, 82, None, "k_rows_mr<double, 20, 2, 2, true, 50, 0>"),
            (1750, 70, None, "k_rows_mr<double, 7, 2, 2, true, 50, 0>"), 
            # the run-time-split wave grids (1344 ... 3840): packed sub-rows with the count at run time (odd: S = 0, even: S = -2) for the centred
            # windows, else staged onto the split rows of fmc_mrfft.h on 64 lanes
            (1344, 82, None, "k_rows_pks<double, -1, 0, 2>"), (1728, 96, None, "k_rows_pks<double, -1, 0, 2>"),
            (1920, 82, None, "k_rows_pks<double, 0, 0, 2>"), (2688, 40, 1310, "k_rows_pks<double, 0, 0, 2>"), (3456, 82, None, "k_rows_pks<double, 0, 0, 2>"),
            (2304, 82, None, "k_rows_pks<double, 1, 0, 2>"), (2560, 82, None, "k_rows_pks<double, 1, -2, 2>"), (3072, 96, None, "k_rows_pks<double, 1, -2, 2>"),
            (3584, 82, 1750, "k_rows_pks<double, 1, -2, 2>"), (3840, 82, None, "k_rows_pks<double, 1, 0, 2>"),
            (2560, 120, None, "k_rows_pks<double, 1, -2, 2, 8>"), (2560, 130, None, "k_rows_pks<double, 1, -2, 2, 16>"), (2560, 130, 200, "k_rows_mr<double, 20, 4, 1, true, 64, 0>"), (1344, 82, 100, "k_rows_mr<double, 7, 2, 1, true, 64, 0>"),
            # ... and of every other multiple of 64 (fmc_core.h: pks_rt): grids whose host-coefficient rows are the chirp-z family's (704 ... 3968)
            # or the 50-lane family's (1600, 3200); off the centred windows the draws are staged onto those rows
            (704, 82, None, "k_rows_pks<double, -1, 0, 2>"), (960, 96, None, "k_rows_pks<double, -1, 0, 2>"), (2112, 82, None, "k_rows_pks<double, -1, 0, 2>"),
            (1408, 82, None, "k_rows_pks<double, 0, 0, 2>"), (3968, 60, 1950, "k_rows_pks<double, 0, 0, 2>"), (2816, 82, None, "k_rows_pks<double, 1, 0, 2>"),
            (1600, 82, None, "k_rows_pks<double, -1, 0, 2>"), (3200, 82, None, "k_rows_pks<double, 0, 0, 2>"),
            (2240, 82, None, "k_rows_pks<double, -1, 0, 2>"), (4032, 96, None, "k_rows_pks<double, -1, 0, 2>"),      # 35 / 63 sub-rows of 64 points
            (832, 120, None, "k_rows_pks<double, -1, 0, 2, 8>"), (832, 120, 300, "k_rows_pbz<double, 8, 1>"), (1600, 82, 3, "k_rows_mr<double, 16, 2, 1, true, 50, 0>"),
            # chirp-z family (any other N): one transform of length 64 P, and rows in input blocks beyond 2048
            # (round 6: windows of up to 128 pixels on the PACKED 256-point pipeline, four rows per wavefront in blocks of 128 inputs -- 6 / 8 planes
            # of the inverse transform; wider windows on the one-row-per-wave chirp-z rows)
            (164, 60, None, "k_rows_pbz<double, 6, 2>"), (291, 82, 5, "k_rows_pbz<double, 6, 2>"),
            (722, 200, None, "k_rows_blu<double, 16, 4, 2, false>"), (1111, 82, None, "k_rows_pbz<double, 6, 2>"),
            (502, 128, 0, "k_rows_pbz<double, 8, 2>"), (1650, 96, 3, "k_rows_pbz<double, 6, 2>"), (129, 97, 32, "k_rows_pbz<double, 8, 2>"),
            (1901, 82, None, "k_rows_pbz<double, 6, 2>"), (3901, 82, None, "k_rows_pbz<double, 6, 2>")]


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
    h = f32_draw_handle(N, Np, "f64", 0)
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


@pytest.mark.parametrize("N,Np,kernel", [(1280, 82, "k_rows_pks<double, 1, 5, 2>"), (896, 60, "k_rows_pks<double, 0, 7, 2>"), (1536, 96, "k_rows_pks<double, 1, 6, 2>"),
                                         (576, 82, "k_rows_pks<double, -1, 9, 2>")])
def test_device_generator_screens_on_the_packed_subrow_grids(N, Np, kernel):
    """`fastmc_screens` (EPI 1: the cropped screens themselves) on the grids of the packed sub-rows: the rows are k_rows_pks, the column
    pass the one-row-per-wave kernel launched alone (dispatch mode -1).  Against the oracle's transform of the restated float64 draws,
    and against the direct family on the same seed."""
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    ps = ps * 0.02
    lo = (N - Np) // 2
    h = f32_draw_handle(N, Np, "f64", 0)
    h.set_spectrum(ps, df)
    h.set_pupil(_window_W(Np), lo, 0.01)
    h.set_rng_precision("f64")
    scr = h.screens(5, 11, 1)
    assert h.last_kernels()[0] == kernel
    z = R.screens_fftw(devrng.device_coefficients_f64(5, 11, N) * (np.sqrt(ps) * df), 1.0)[lo:lo + Np, lo:lo + Np]
    assert np.abs(scr[0] - z.real).max() < 1e-11 * np.abs(z).max() and np.abs(scr[1] - z.imag).max() < 1e-11 * np.abs(z).max()
    h.kernel_path(0)
    np.testing.assert_allclose(h.screens(5, 11, 1), scr, rtol=0, atol=1e-11 * np.abs(z).max())


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


def test_wide_windows_on_split_grids_use_the_wave_family():
    """Windows of 257-512 pixels (a 2.6-5 m aperture at 1 cm) at 2048^2 stay on the wave kernels (eight output slots per
    lane) with the device generator, and agree with the direct family."""
    h, ps, df, W = _small_problem(2048, 402)
    assert h.kernel_path() == 1
    a = h.run(5, 0, 2, None, 0.01)
    assert h.last_timing()["rows_launches"] == 1 and np.isfinite(a).all()
    h.kernel_path(0)
    np.testing.assert_allclose(h.run(5, 0, 2, None, 0.01), a, rtol=1e-9)


def test_dense_sixteen_wave_kernels_equal_the_twelve_wave_kernels():
    """1024^2 with a window of up to 96 pixels runs the dense-image kernels (sixteen waves per workgroup); with
    FASTMC_NO_DENSE16=1 the same library keeps the twelve-wave kernels: same arithmetic, same results."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from _parity import _small_problem; "
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


@pytest.mark.parametrize("N,n", [(1024, 420), (2048, 320)])
def test_tile_walk_equals_one_tile_per_workgroup(N, n):
    """A launch of at least eight rounds of workgroups keeps as many workgroups as the device holds and lets them walk the launch's
    tiles (k_rows_wave: RowArgs::tiles; k_cols_wave of the 1024-point pipeline: each wave walks the columns); FASTMC_ROWS_PERSIST=0 /
    FASTMC_COLS_PERSIST=0 give every tile a workgroup of its own.  Same arithmetic per row and column, partial sums per column:
    the results are the same BITS, with both generators."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from _parity import _small_problem; "
            "h, ps, df, W = _small_problem(%d, 82); a = h.run(23, 1, %d, None, 0.02); h.set_rng_precision('f64'); "
            "np.save(sys.argv[1], np.stack([a, h.run(23, 1, %d, None, 0.02)]))") % (ROOT, N, n, n)
    outs = []
    for flag in ("0", "1"):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fastmc_walk_{flag}_{os.getpid()}.npy")
        env = dict(os.environ, FASTMC_ROWS_PERSIST=flag, FASTMC_COLS_PERSIST=flag, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
        os.remove(path)
    assert np.isfinite(outs[0]).all() and outs[0].shape[0] == 2 and not np.array_equal(outs[0][0], outs[0][1])
    np.testing.assert_array_equal(outs[0], outs[1])


# ------------------------------------------------------------------ grids beyond 4096 (the reference has no upper limit, fast.py:176-211)
@pytest.mark.parametrize("N,Np,kernel,kernel64", [
    # multiples of 64: the packed sub-rows with a run-time count for the centred windows (18 / 32 sub-rows of 256 points, 65 of 64) ...
    (4608, 60, "k_rows_pks<double, 1, -2, 0>", "k_rows_pks<double, 1, -2, 2>"), (8192, 82, "k_rows_pks<double, 1, -2, 0>", "k_rows_pks<double, 1, -2, 2>"),
    (4160, 82, "k_rows_pks<double, -1, 0, 0>", "k_rows_pks<double, -1, 0, 2>"),
    # ... any other window: the float32 draw on the direct family, the float64 generator staged onto the family's host-coefficient rows
    (7168, 100, "k_rows_pks<double, 1, -2, 0, 8>", "k_rows_pks<double, 1, -2, 2, 8>"),
    (6144, 140, "k_rows_pks<double, 1, -2, 0, 16>", "k_rows_pks<double, 1, -2, 2, 16>"),
    (5000, 82, "k_rows_mr<double, 20, 2, 0, true, 50, 1>", "k_rows_mr<double, 20, 2, 2, true, 50, 0>"),
    (4100, 82, "k_rows_pbz<double, 6, 0>", "k_rows_pbz<double, 6, 2>"),
    (7003, 200, "k_rows_blu<double, 16, 4, 0, true>", "k_rows_blu<double, 16, 4, 2, true>")])
def test_grids_beyond_4096(N, Np, kernel, kernel64):
    """Up to 8192: multiples of 64 on the packed sub-rows (fmc_core.h: pks_rt), N = 64 P S / 50 P S with a run-time sub-row count S <= 8
    (wave_rt_split / mr_split) for their host coefficients and other windows, and any other N on the chirp-z kernels with its rows in up
    to eleven input blocks: screens from host coefficients against numpy's FFT; the device generator through the rows against the
    oracle on the restated draws (one size: the restatement is Python) and against the direct family on the same seed; the float64
    generator against the direct family's staged draws."""
    rng = np.random.default_rng(N)
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    ps = ps * 0.02
    lo = (N - Np) // 2
    W = _window_W(Np)
    h = f32_draw_handle(N, Np, "f64", 0)
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
    assert h.last_kernels()[0] == kernel64
    h.kernel_path(0)
    np.testing.assert_allclose(got64, h.run(seed, real0, 1, None, 0.01), rtol=1e-9)
    h.set_rng_precision("f32")
    np.testing.assert_allclose(got, h.run(seed, real0, 1, None, 0.01), rtol=1e-9)
    h.close()


def test_handle_default_generator_is_the_reference_precision():
    """Round 5: `fastmc_create` leaves a handle drawing at the precision it computes in -- a float64 handle draws the reference's
    53-bit normals (fast/funcs.py:352-356), fused into its row kernel, with no call to fastmc_set_rng_precision; a float32
    handle the float32 draw.  The state survives the handle cache (a parked handle comes back reset to it)."""
    N, Np = 1024, 82
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    for round_ in range(2):                      # second pass: the handle parked by close() is taken back by fastmc_create
        h = _lib.Handle(N, Np, "f64", 0)
        got = h.rng_coeffs(3, 1)
        assert np.abs(got - devrng.device_coefficients_f64(3, 1, N)).max() < 2e-14
        h.set_spectrum(ps * 0.02, df)
        h.set_pupil(_window_W(Np), (N - Np) // 2, 0.01)
        h.run(3, 0, 2, None, 0.01)
        assert h.last_kernels()[0] == "k_rows_wave<double, 16, 2, 2, 1, 4>"
        h.set_rng_precision("f32")               # the opt-in: must not leak into the next handle of this shape
        assert np.abs(h.rng_coeffs(3, 1) - devrng.device_coefficients(3, 1, N)).max() < 1e-3
        h.close()
    h32 = _lib.Handle(N, Np, "f32", 0)
    assert h32.precision == "f32"
    assert np.abs(h32.rng_coeffs(3, 1) - devrng.device_coefficients(3, 1, N)).max() < 1e-3
    assert np.abs(h32.rng_coeffs(3, 1) - devrng.device_coefficients_f64(3, 1, N)).max() > 1e-9
    h32.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,Np", [(256, 82), (2048, 82), (512, 200)])
def test_a_forced_family_keeps_the_grids_generator_layout(N, Np):
    """`fastmc_kernel_path` can force the chirp-z family onto a grid of the wave family whose generator layout is not the 64 streams
    per row the chirp-z rows draw (packed grids: N / 16; 2048: 128).  Results depend on (seed, realisation) only, never on the kernel
    family: the forced family must take the same draws (float64 generator staged, float32 draw on the direct family)."""
    ps, df = _vk_spectrum(N, 0.01, 30.0)
    ps = ps * 0.02
    lo = (N - Np) // 2
    h = f32_draw_handle(N, Np, "f64", 0)
    h.set_spectrum(ps, df)
    h.set_pupil(_window_W(Np), lo, 0.01)
    for prec in ("f64", "f32"):
        h.kernel_path(1)
        h.set_rng_precision(prec)
        want = h.run(11, 5, 2, None, 0.01)
        assert h.kernel_path(2) == 2
        got = h.run(11, 5, 2, None, 0.01)
        assert "blu" in h.last_kernels()[1] or "pbz" in h.last_kernels()[1] or "direct" in h.last_kernels()[1]
        np.testing.assert_allclose(got, want, rtol=1e-9 if prec == "f64" else 2e-6)
    h.close()
