"""GPU parity, the reference's own outputs: same-seed end-to-end fixtures at small and full size, the power-spectrum kernel and its
terms, TEMPORAL, masks, mean irradiance, the object's attributes -- everything compared with tests/golden/ (captured from /root/reference)."""
from _parity import *      # noqa: F401,F403 (numpy, pytest, fixtures, fast_amd, the oracle, the shared helpers)

pytestmark = pytest.mark.gpu


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
    h = f32_draw_handle(N, Np, prec, 0)
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
    h = f32_draw_handle(N, Np, "f64", 0)
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


@pytest.mark.parametrize("case", ["default164", "oddN", "oddNp", "autosize", "subharm"])
def test_fast_run_reproduces_reference_on_chirpz_kernels(case):
    """The reference's own grids (auto-sized 164, odd 49, 48, 64) forced onto the chirp-z family: same SEED -> same `_r`."""
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "host", "GPU_DEVICE": 0, "GPU_KERNELS": "chirpz"})
    sim = fast_amd.Fast(p)
    assert sim._handle.kernel_path() == 2
    np.testing.assert_allclose(sim.run()._r, g["r"], rtol=1e-9)


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


@pytest.mark.parametrize("name", ["e2e_explicit_pupil", "e2e_explicit_pupil_noao"])
@pytest.mark.parametrize("rng_mode", ["host", "numpy"])
def test_stand_in_free_fixtures_same_seed(name, rng_mode):
    """rows a5 / a10 pinned by reference runs with NO aotools stand-in in the chain -- the pupil weights are
    explicit arrays assigned to the object (as a user of the reference would), the pupil filter is 1; the device's power
    spectrum, Simpson scalars, log-amplitudes, last chunk of screens and `result._r` against the reference's, with numpy's
    draws uploaded (GPU_RNG 'host') and with numpy's stream drawn on the device ('numpy')."""
    from conftest import run_with_explicit_pupil, check_explicit_pupil_run
    g, sim, res = run_with_explicit_pupil(name, GPU_DEVICE=0, GPU_RNG=rng_mode)
    check_explicit_pupil_run(g, sim, res)


def test_detector_kat_with_explicit_weights():
    """The detector known-answer of test_detector_kat_from_reference with weights that were written out in numpy."""
    g = load_golden("kat_detector_explicit_W")
    phs, W, M = g["phs"], g["W"], int(g["M"])
    Np, N = W.shape[0], 64
    la = g["logamp"][int(g["chunk"]) * M:(int(g["chunk"]) + 1) * M]
    screen = np.zeros((1, N, N))
    corners = [(0, 0), (0, Np + 3), (Np + 5, 1), (Np + 7, Np + 9)]
    for j, (r0, c0) in enumerate(corners):
        screen[0, r0:r0 + Np, c0:c0 + Np] = phs[j]
    xs = np.stack([r0 + np.arange(Np, dtype=float) for r0, _ in corners])[None]
    ys = np.stack([c0 + np.arange(Np, dtype=float) for _, c0 in corners])[None]
    h = f32_draw_handle(N, Np, "f64", 0)
    h.set_pupil(W, (N - Np) // 2, float(g["dx"]))
    h.set_layer_screens(screen)
    inc = h.temporal_chunk(xs, ys, np.zeros((1, 2, M), dtype=np.int32), la, coherent=False)
    coh = h.temporal_chunk(xs, ys, np.zeros((1, 2, M), dtype=np.int32), la, coherent=True)
    np.testing.assert_allclose(inc, g["incoherent"], rtol=1e-12)
    np.testing.assert_allclose(coh, g["coherent"], rtol=1e-12, atol=1e-15)
