"""GPU parity, results and plumbing: histogram, statistics and link metrics on the resident results, errors, several handles and
threads, BASELINE configs[3] and [4] at full size, the world-of-one RCCL exchange."""
from _parity import *      # noqa: F401,F403 (numpy, pytest, fixtures, fast_amd, the oracle, the shared helpers)

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ properties at BASELINE size
def test_full_size_properties_1024():
    N, Np = 1024, 82
    ps, df = _vk_spectrum(N, 0.01, 25.0)
    W = _window_W(Np)
    h = f32_draw_handle(N, Np, "f64", 0)
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
    h = f32_draw_handle(64, 22, "f64", 0)
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


def test_last_clock_never_reports_the_stamps_of_an_earlier_launch():
    """only k_rows_wave stamps the clock.  After another kernel family ran last on the handle, fastmc_last_clock says so
    (None here) instead of handing out the previous launch's stamps; the next stamping launch brings it back."""
    from fast_amd import _lib
    h = _lib.Handle(1024, 82, "f64", 0)
    h.set_spectrum(np.ones((1024, 1024)) * 1e-4, 0.5)
    h.set_pupil(np.ones((82, 82)), (1024 - 82) // 2, 0.01)
    assert h.last_clock() is None
    h.run(1, 0, 64, None, 0.01)
    ghz, span = h.last_clock()
    assert 1.2 < ghz < 2.5 and span > 1
    h.kernel_path(0)                      # the direct family: no stamps
    h.run(1, 0, 2, None, 0.01)
    assert h.last_kernels()[0].startswith("k_rows_direct") and h.last_clock() is None
    h.kernel_path(1)
    h.run(1, 0, 64, None, 0.01)
    assert h.last_clock() is not None
    h.close()


def test_zenith_scan_dealt_over_devices_equals_the_single_device_scan():
    """`sweep.zenith_scan(devices=[...])` deals a process's samples to one thread per device (bench.py --gpus N
    without a launcher: BASELINE configs[4] is 4 configs per GPU on 8 GPUs).  With GPU_DEVICES=[0, 0, 0, 0] on the one GPU of a
    test box every record -- the per-iteration powers too -- equals the single-device scan's, and each worker served its share."""
    from fast_amd import sweep
    g = load_golden("e2e_ao_alias")
    p = params_from_json(g["params_json"])
    p.update({"GPU_DEVICE": 0, "NPXLS": 512, "SEED": 3})
    angles = np.linspace(0, 60, 10)
    one = sweep.zenith_scan(p, angles, niter=64, keep_power=True)
    four = sweep.zenith_scan(p, angles, niter=64, keep_power=True, devices=[0, 0, 0, 0])
    assert [r["index"] for r in four] == list(range(10)) and all(r["device"] == 0 for r in four)
    for a, b in zip(one, four):
        assert a["zenith"] == b["zenith"] and a["phs_var"] == b["phs_var"] and a["mean_dB_rel"] == b["mean_dB_rel"]
        np.testing.assert_array_equal(a["r"], b["r"])
    # ... and combined with a rank's share of a multi-process launch: rank 1 of 2 owns the odd samples, dealt over two devices
    half = sweep.zenith_scan(p, angles, niter=64, rank=1, world=2, devices=[0, 0])
    assert [r["index"] for r in half] == [1, 3, 5, 7, 9] and half[2]["mean_dB_rel"] == one[5]["mean_dB_rel"]
    # a worker's exception reaches the caller
    with pytest.raises(Exception):
        sweep.zenith_scan(dict(p, NPXLS=-5), angles[:4], niter=64, devices=[0, 0])


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
    h = f32_draw_handle(N, Np, "f32", 0)
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
