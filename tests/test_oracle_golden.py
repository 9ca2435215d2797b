"""Pins oracle/fastref.py against outputs of the reference itself (tests/golden)."""
import numpy as np
import pytest

from conftest import E2E_CASES, load_golden, params_from_json
from oracle import fastref as R


@pytest.mark.parametrize("N", [16, 30, 33, 64, 128, 100, 150])
def test_fft_branch_matches_reference(N):
    g = load_golden(f"kat_fft_N{N}")
    z = R.screens_fftw(g["coeffs"] * np.sqrt(g["powerspec"]), float(g["df"]))
    scr = R.double_screens(z)
    scale = np.abs(g["screens"]).max()
    assert np.abs(scr - g["screens"]).max() <= 1e-13 * scale


def test_detector_matches_reference():
    g = load_golden("kat_detector")
    M, c = int(g["M"]), int(g["chunk"])
    la = g["logamp"][c * M:(c + 1) * M]
    np.testing.assert_allclose(R.detector(g["phs"], g["W"], float(g["dx"]), la), g["incoherent"], rtol=1e-13)
    np.testing.assert_allclose(R.detector(g["phs"], g["W"], float(g["dx"]), la, coherent=True),
                               g["coherent"], rtol=1e-13)


def test_von_karman_and_simpson():
    g = load_golden("kat_vk")
    grid = R.main_grid(int(g["N"]), float(g["dx"]))
    np.testing.assert_allclose(R.von_karman(grid.fabs, g["cn2"], np.inf, 1e-6), g["vk_inf"], rtol=1e-14)
    vk = R.von_karman(grid.fabs, g["cn2"], 25.0, 0.01)
    np.testing.assert_allclose(vk, g["vk_L0"], rtol=1e-14)
    np.testing.assert_allclose(R.simpson2d(vk, grid.axis_x), g["simpson_vk_L0"], rtol=1e-14)
    assert g["vk_inf"][:, 12, 12].tolist() == [0.0, 0.0, 0.0]


def test_subharm_screens():
    g = load_golden("kat_subharm")
    sh = R.subharm_grid(int(g["N"]), float(g["dx"]))
    np.testing.assert_allclose(sh.fx, g["fx"], rtol=1e-15)
    np.testing.assert_allclose(sh.fy, g["fy"], rtol=1e-15)
    np.testing.assert_allclose(sh.df, g["df"], rtol=1e-15)
    scr = R.subharm_screens(g["rand"], sh, int(g["N"]), float(g["dx"]))
    np.testing.assert_allclose(scr, g["screens"], rtol=1e-10, atol=1e-12 * np.abs(g["screens"]).max())


def test_masks():
    g = load_golden("kat_masks")
    grid = R.main_grid(int(g["N"]), float(g["dx"]))
    assert np.array_equal(R.mask_lf(grid.fx, grid.fy, 0.08), g["zonal"])
    assert np.array_equal(R.mask_lf(grid.fx, grid.fy, 0.08, modal=True, modal_mult=0.7), g["modal"])
    np.testing.assert_allclose(R.mask_lf(grid.fx, grid.fy, 0.08, modal=True, zmax=3, D=0.4), g["zern3"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(R.mask_lf(grid.fx, grid.fy, 0.08, modal=True, zmax=9, D=0.4), g["zern9"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(R.zernike_sq_filter(grid.fabs, grid.fx, grid.fy, 0.4, 4), g["zsq4"], rtol=1e-12, atol=1e-15)


def test_rng_anchor():
    g = load_golden("kat_turbulence")
    assert np.array_equal(np.random.default_rng(1).normal(0, 1, 3), g["rng_anchor"])


def _spec_inputs(g):
    p = params_from_json(g["params_json"])
    ao_mode = p["AO_MODE"]
    zmax = p["ZMAX"]
    modal, mult = p["MODAL"], p["MODAL_MULT"]
    if ao_mode == "TT":
        zmax, modal, mult = 3, True, 1
    return p, dict(N=int(g["Npxls"]), dx=float(g["dx"]), cn2=g["cn2"], h=g["h"], wind=g["wind_vector"],
                   L0=p["L0"], l0=p["l0"], wvl=p["WVL"], ao_mode=ao_mode, d_wfs=p["DSUBAP"],
                   dtheta=p["DTHETA"], D_ground=p["D_GROUND"], zmax=zmax, t_loop=p["TLOOP"],
                   t_exp=p["TEXP"], alias=p["ALIAS"], noise=p["NOISE"]), (modal, mult)


@pytest.mark.parametrize("case", E2E_CASES + ["default164"])
def test_residual_powerspec_matches_reference(case):
    g = load_golden("e2e_" + case)
    p, kw, (modal, mult) = _spec_inputs(g)
    grid = R.main_grid(kw["N"], kw["dx"])
    mask = R.mask_lf(grid.fx, grid.fy, kw["d_wfs"], modal=modal, modal_mult=mult, zmax=kw["zmax"], D=kw["D_ground"])
    np.testing.assert_allclose(mask, g["lf_mask"], rtol=1e-12, atol=1e-15)
    out = R.residual_powerspec(lf_mask=g["lf_mask"], pupil_filter=g["pupil_filter"], **kw)
    tiny = 1e-13 * np.abs(g["powerspec"]).max()
    np.testing.assert_allclose(out["powerspec"], g["powerspec"], rtol=1e-11, atol=tiny)
    np.testing.assert_allclose(out["powerspec_per_layer"], g["powerspec_per_layer"], rtol=1e-11, atol=tiny)
    np.testing.assert_allclose(out["logamp_powerspec"], g["logamp_powerspec"], rtol=1e-11,
                               atol=1e-13 * np.abs(g["logamp_powerspec"]).max())
    for k in ("logamp_var", "phs_var", "fitting_error", "aniso_servo_error", "alias_error", "noise_error"):
        np.testing.assert_allclose(out[k], g[k], rtol=1e-10, atol=1e-300, err_msg=k)
    np.testing.assert_allclose(out["phs_var_weights"], g["phs_var_weights"], rtol=1e-10)


@pytest.mark.parametrize("case", E2E_CASES + ["default164"])
def test_monte_carlo_matches_reference_same_seed(case):
    """Identical numpy seed -> identical `result._r` (draw order restated exactly)."""
    g = load_golden("e2e_" + case)
    p, kw, (modal, mult) = _spec_inputs(g)
    N, dx = kw["N"], kw["dx"]
    W = g["pupil"] * g["pupil_mode"]
    sub = None
    if p["SUBHARM"]:
        ps_lo, sh = R.subharm_powerspec(modal=modal, modal_mult=mult, **kw)
        np.testing.assert_allclose(ps_lo, g["powerspec_subharm"], rtol=1e-11)
        sub = (ps_lo, sh)
    r = R.monte_carlo(p["SEED"], p["NITER"], p["NCHUNKS"], g["powerspec"], R.main_grid(N, dx).df, W, dx,
                      float(g["logamp_var"]), coherent=p["COHERENT"], sub=sub)
    np.testing.assert_allclose(r, g["r"], rtol=1e-10)
    assert r.dtype == g["r"].dtype


# ------------------------------------------------------------------ temporal (frozen-flow) mode
@pytest.mark.parametrize("name", ["temporal_default", "temporal_small", "temporal_noao", "temporal_npxls100"])
def test_temporal_mode_matches_reference(name):
    g = load_golden(name)
    p = params_from_json(g["params_json"])
    N, Np, dx = int(g["Npxls"]), int(g["Npxls_pup"]), float(g["dx"])
    grid = R.main_grid(N, dx)
    L = len(g["h"])
    ax, ay, fabs = R.temporal_freqs(L, N, p["NITER"], g["wind_speed"], g["wind_dir"], p["DT"], grid.df)
    np.testing.assert_allclose(ax, g["fx_axis_t"], rtol=1e-14)
    np.testing.assert_allclose(fabs, g["fabs_t"], rtol=1e-13, atol=1e-12)
    spline = R.temporal_pupil_filter(ax, ay, grid.df, p["D_GROUND"], p["OBSC_GROUND"], float(g["W0"]), Np, dx)
    tps = R.temporal_logamp_spectrum(ax, ay, fabs, g["h"], g["cn2"], p["WVL"], spline, p["L0"], p["l0"], grid.df)
    np.testing.assert_allclose(tps, g["temporal_logamp_powerspec"], rtol=1e-9, atol=1e-30)
    np.testing.assert_allclose(R.temporal_pixel_shifts(p["NITER"] // p["NCHUNKS"], p["DT"], g["wind_vector"], dx),
                               g["pixel_shifts"], rtol=1e-14)
    W = g["pupil"] * g["pupil_mode"]
    r = R.monte_carlo_temporal(p["SEED"], p["NITER"], p["NCHUNKS"], g["powerspec_per_layer"], grid.df, W, dx,
                               float(g["logamp_var"]), g["temporal_logamp_powerspec"], g["wind_vector"], p["DT"], N, Np,
                               coherent=p["COHERENT"])
    np.testing.assert_allclose(r, g["r"], rtol=1e-9)


def test_link_metrics_match_reference():
    """oracle fade / BER restatement vs reference fast/comms.py:171-262 (comms_metrics.npz)."""
    d = load_golden("comms_metrics")
    thr, eb, Ms, dt = d["thresholds"], d["ebn0"], d["Ms"], float(d["dt"])
    for n in d["names"]:
        v = d["v_" + n]
        np.testing.assert_array_equal([R.fade_prob(v, t) for t in thr], d["fade_prob_" + n])
        np.testing.assert_array_equal([R.fade_prob(v, t, 5) for t in thr], d["fade_prob_min5_" + n])
        np.testing.assert_allclose([R.fade_dur(v, t, dt) for t in thr], d["fade_dur_" + n], rtol=1e-14)
        np.testing.assert_allclose([R.fade_dur(v, t, dt, 5) for t in thr], d["fade_dur_min5_" + n], rtol=1e-14)
        np.testing.assert_allclose([R.ber_ook(s, v) for s in eb], d["ber_ook_" + n], rtol=1e-13)
        np.testing.assert_allclose([[R.sep_qam(M, s, v) for s in eb] for M in Ms], d["sep_qam_" + n], rtol=1e-13)
        np.testing.assert_allclose([[R.ber_qam(M, s, v) for s in eb] for M in Ms], d["ber_qam_" + n], rtol=1e-13)
    np.testing.assert_allclose([R.ber_ook(s) for s in eb], d["ber_ook_nosamples"], rtol=1e-14)
    np.testing.assert_allclose([[R.sep_qam(M, s) for s in eb] for M in Ms], d["sep_qam_nosamples"], rtol=1e-14)
    np.testing.assert_allclose([[R.ber_qam(M, s) for s in eb] for M in Ms], d["ber_qam_nosamples"], rtol=1e-14)


def test_mean_irradiance_matches_reference():
    """oracle restatement of Fast.compute_mean_irradiance (fast.py:736-761) vs the captured outputs."""
    g = load_golden("mean_irradiance")
    on = R.mean_irradiance(g["powerspec"], g["W"], float(g["dx"]), float(g["df"]), float(g["diffraction_limit"]))
    np.testing.assert_allclose(on, g["onaxis"], rtol=1e-10)
    off = R.mean_irradiance(g["powerspec"], g["W"], float(g["dx"]), float(g["df"]), float(g["diffraction_limit"]), onaxis=False)
    np.testing.assert_allclose(off, g["offaxis"], rtol=1e-9, atol=1e-12 * np.abs(g["offaxis"]).max())
    on2 = R.mean_irradiance(g["powerspec2"], g["W2"], float(g["dx2"]), float(g["df2"]), float(g["diffraction_limit2"]))
    np.testing.assert_allclose(on2, g["onaxis2"], rtol=1e-10)


def test_default_numpy_branch_fixture_and_its_relation_to_the_fftw_branch():
    """The reference's DEFAULT transform branch (FFTW False, funcs.py:216-218: aotools ift2) next to the FFTW branch
    the GPU computes, same SEED (fixture e2e_numpy_branch; STAND-IN DEPENDENT in its arithmetic: ift2 is our
    stand-in of aotools).  (i) the oracle's restatement of that branch reproduces the reference's screens; (ii) per
    chunk of B complex transforms on an N x N grid the two branches are tied by
        numpy[b] = (B / N)^2 * point_mirror(FFTW[(b + 2 (B // 2)) mod B])      (mirror: index -> (N - index) mod N)
    i.e. the default branch sees the same screens mirrored, rolled along the chunk axis (all-axes ifftshift) and
    scaled by (chunk / N)^2 (`N = DATA.shape[0]` inside ift2 is the chunk length): with chunk != N they are NOT the
    same physical phase.  fast_amd always computes the FFTW branch and warns when FFTW is False."""
    g = load_golden("e2e_numpy_branch")
    p = params_from_json(g["params_json"])
    N, Np = int(g["Npxls"]), int(g["Npxls_pup"])
    B = p["NITER"] // p["NCHUNKS"] // 2
    rng = np.random.default_rng(p["SEED"])
    R.draw_logamp(rng, p["NITER"], float(g["logamp_var"]))
    for _ in range(p["NCHUNKS"]):
        coeffs = R.draw_coefficients(rng, (B, N, N))                          # last chunk's draws
    col = coeffs * np.sqrt(g["powerspec"])
    npb = R.crop(R.double_screens(R.screens_numpy_branch(col, float(g["df"]))), N, Np)
    np.testing.assert_allclose(npb, g["phs_last_chunk"], rtol=1e-12, atol=1e-14 * np.abs(g["phs_last_chunk"]).max())
    fw = R.crop(R.double_screens(R.screens_fftw(col, float(g["df"]))), N, Np)
    np.testing.assert_allclose(fw, g["phs_last_chunk_fftw"], rtol=1e-12, atol=1e-14 * np.abs(fw).max())
    # the relation, on the full grid
    zf = R.screens_fftw(col, float(g["df"]))
    zn = R.screens_numpy_branch(col, float(g["df"]))
    idx = (N - np.arange(N)) % N
    mirrored = zf[:, idx][:, :, idx]
    want = (B / N) ** 2 * np.roll(mirrored, -2 * (B // 2), axis=0)
    np.testing.assert_allclose(zn, want, rtol=1e-10, atol=1e-13 * np.abs(want).max())
    # consequence for the results: the default branch's phases are (B/N)^2 smaller -> powers near the no-turbulence value
    assert g["r"].mean() > g["r_fftw"].mean()


# ------------------------------------------------------------------ fixtures with no aotools stand-in in the chain
def test_detector_with_explicit_weights():
    """SURVEY 8 row a5 pinned with NOTHING of ours in the fixture's making: the weights were written out in numpy by the capture
    script (tools/capture_golden/capture.py: explicit_weights), the phase cube is random, the arithmetic is the reference's."""
    g = load_golden("kat_detector_explicit_W")
    M, c = int(g["M"]), int(g["chunk"])
    la = g["logamp"][c * M:(c + 1) * M]
    np.testing.assert_allclose(R.detector(g["phs"], g["W"], float(g["dx"]), la), g["incoherent"], rtol=1e-13)
    np.testing.assert_allclose(R.detector(g["phs"], g["W"], float(g["dx"]), la, coherent=True), g["coherent"], rtol=1e-13)


@pytest.mark.parametrize("name", ["e2e_explicit_pupil", "e2e_explicit_pupil_noao"])
def test_end_to_end_with_explicit_pupil_weights(name):
    """Rows a6-a10 and a1-a5b end to end: residual spectrum, log-amplitude variance (pupil_filter = 1) and the same-seed powers
    against a reference run whose pupil, fibre mode and pupil filter were explicit arrays."""
    g = load_golden(name)
    p = params_from_json(g["params_json"])
    N, dx = int(g["Npxls"]), float(g["dx"])
    ao_mode = p["AO_MODE"]
    out = R.residual_powerspec(N=N, dx=dx, cn2=g["cn2"], h=g["h"], wind=g["wind_vector"], L0=p["L0"], l0=p["l0"], wvl=p["WVL"],
                               ao_mode=ao_mode, d_wfs=p["DSUBAP"], dtheta=p["DTHETA"], D_ground=p["D_GROUND"], zmax=None,
                               t_loop=p["TLOOP"], t_exp=p["TEXP"], alias=p["ALIAS"], noise=p["NOISE"], lf_mask=g["lf_mask"],
                               pupil_filter=np.ones((N, N)))
    tiny = 1e-13 * np.abs(g["powerspec"]).max()
    np.testing.assert_allclose(out["powerspec"], g["powerspec"], rtol=1e-11, atol=tiny)
    np.testing.assert_allclose(out["logamp_powerspec"], g["logamp_powerspec"], rtol=1e-11, atol=1e-13 * np.abs(g["logamp_powerspec"]).max())
    for k in ("logamp_var", "phs_var", "fitting_error", "aniso_servo_error", "alias_error", "noise_error"):
        np.testing.assert_allclose(out[k], g[k], rtol=1e-10, atol=1e-300, err_msg=k)
    r = R.monte_carlo(p["SEED"], p["NITER"], p["NCHUNKS"], g["powerspec"], float(g["df"]), g["W"], dx, float(g["logamp_var"]),
                      coherent=p["COHERENT"])
    np.testing.assert_allclose(r, g["r"], rtol=1e-10)
    assert r.dtype == g["r"].dtype
