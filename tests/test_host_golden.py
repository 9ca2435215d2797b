"""CPU: fast_amd's host-side init (fast_amd/host.py) against the reference's init products."""
import numpy as np
import pytest

from conftest import E2E_CASES, load_golden, params_from_json
import fast_amd
from fast_amd import host, hostmath, turbulence_models as tm


def _problem(case):
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    c = fast_amd.conf.ConfigParser(dict(p))
    return g, c.config, host.build_problem(c.config)


@pytest.mark.parametrize("case", E2E_CASES + ["default164"])
def test_host_init_matches_reference(case):
    g, p, prob = _problem(case)
    atm, pup = prob.atm, prob.pup
    assert prob.N == int(g["Npxls"]) and prob.Np == int(g["Npxls_pup"])
    for name, val in (("dx", prob.dx), ("L", atm.L), ("paa", atm.paa), ("r0", atm.r0), ("theta0", atm.theta0),
                      ("tau0", atm.tau0), ("r0_los", atm.r0_los), ("theta0_los", atm.theta0_los),
                      ("tau0_los", atm.tau0_los), ("zenith_correction", atm.zenith_correction), ("k", prob.k),
                      ("dx_sat", pup.dx_sat), ("W0", pup.W0), ("W0_sat", pup.W0_sat),
                      ("diffraction_limit", prob.diffraction_limit)):
        np.testing.assert_allclose(val, g[name], rtol=1e-12, err_msg=name)
    for name, val in (("h", atm.h), ("cn2", atm.cn2), ("wind_vector", atm.wind_vector), ("wind_speed", atm.wind_speed),
                      ("pupil", pup.pupil), ("pupil_mode", pup.pupil_mode), ("pupil_sat", pup.pupil_sat),
                      ("pupil_mode_sat", pup.pupil_mode_sat)):
        np.testing.assert_allclose(val, g[name], rtol=1e-12, atol=1e-300, err_msg=name)
    np.testing.assert_allclose(pup.pupil_filter, g["pupil_filter"], rtol=1e-9, atol=1e-18)
    fx, fy, _ = host.mesh(prob.axis)
    m = host.lf_mask(fx, fy, prob.d_wfs, prob.modal, prob.modal_mult, prob.zmax, p["D_GROUND"])
    np.testing.assert_allclose(np.asarray(m, dtype=float), g["lf_mask"], rtol=1e-12, atol=1e-15)
    assert np.array_equal(pup.pup_coords, g["pup_coords"])
    keys = [str(k) for k in g["link_budget_keys"]]
    assert keys == list(prob.link_budget.keys())
    np.testing.assert_allclose(list(prob.link_budget.values()), g["link_budget_vals"], rtol=1e-12)


@pytest.mark.parametrize("case", ["subharm", "subharm_ao", "oddN", "npxls150"])
def test_subharm_spectrum_matches_reference(case):
    g, p, prob = _problem(case)
    ps, fx, fy, df, sh = host.subharm_spectrum(prob)
    np.testing.assert_allclose(ps, g["powerspec_subharm"], rtol=1e-11)
    np.testing.assert_allclose(fx, g["sh_fx"], rtol=1e-15)
    np.testing.assert_allclose(fy, g["sh_fy"], rtol=1e-15)
    np.testing.assert_allclose(df, g["sh_df"], rtol=1e-15)
    # the bookkeeping attributes of fast.py:494-526
    np.testing.assert_allclose(sh.per_layer, g["powerspec_subharm_per_layer"], rtol=1e-11)
    np.testing.assert_allclose(sh.phs_var, g["phs_var_subharm"], rtol=1e-11)
    np.testing.assert_allclose(sh.phs_var_weights, g["phs_var_weights_sh"], rtol=1e-11)
    np.testing.assert_allclose(np.asarray(sh.lf_mask, dtype=float), g["lf_mask_subharm"], rtol=1e-12, atol=1e-15)


def test_simpson_weights_reproduce_scipy():
    from scipy.integrate import simpson
    rng = np.random.default_rng(0)
    for n in (16, 23, 64, 164):
        f = host.freq_axis(n, 0.013)
        w = hostmath.simpson_weights(f)
        P = rng.random((3, n, n))
        np.testing.assert_allclose(np.einsum("i,lij,j->l", w, P, w), simpson(simpson(P, x=f), x=f), rtol=1e-12)


def test_turbulence_models():
    g = load_golden("kat_turbulence")
    h4, c4, w4 = tm.HV57_Bufton_profile(4)
    np.testing.assert_allclose(h4, g["h4"], rtol=1e-13)
    np.testing.assert_allclose(c4, g["cn2_4"], rtol=1e-13)
    np.testing.assert_allclose(w4, g["w4"], rtol=1e-13)
    h10, c10, w10 = tm.HV57_Bufton_profile(10, w=30, A=3e-14, vg=5)
    np.testing.assert_allclose(np.r_[h10, c10, w10], np.r_[g["h10"], g["cn2_10"], g["w10"]], rtol=1e-13)
    np.testing.assert_allclose(tm.HV57(g["hh"]), g["hv57"], rtol=1e-14)
    np.testing.assert_allclose(tm.Bufton_wind(g["hh"]), g["bufton"], rtol=1e-14)
    assert len(tm.HV57(g["hh"])) == 10 and tm.HV57(g["hh"]).dtype == float          # reference test_HV57
    np.testing.assert_allclose([host.l_path(36e6, 55.0), host.l_path(600e3, 0.0), host.l_path(600e3, 70.0)], g["l_path"], rtol=1e-14)
    np.testing.assert_allclose(host.wind_correction(np.array([1e3, 1e4]), [30.0, -12.0], 1e-3), g["wind_corr"], rtol=1e-14)


def test_config_parser_defaults_and_errors(tmp_path):
    c = fast_amd.conf.ConfigParser({"NITER": 10})
    assert c.config["NCHUNKS"] == 10 and c.config["AO_MODE"] == "AO" and c.config["GPU_RNG"] == "device"
    cfg = tmp_path / "cfg.py"
    cfg.write_text("p = {'NITER': 4, 'NCHUNKS': 2}\n")
    assert fast_amd.conf.ConfigParser(str(cfg)).config["NITER"] == 4
    with pytest.raises(Exception):
        fast_amd.conf.ConfigParser(str(tmp_path / "cfg.yaml"))
    with pytest.raises(Exception):
        fast_amd.conf.ConfigParser(3)
    base = dict(fast_amd.conf.DEFAULTS)
    with pytest.raises(Exception, match="NCHUNKS must divide"):
        host.build_problem({**base, "NITER": 10, "NCHUNKS": 3})
    with pytest.raises(Exception, match="even"):
        host.build_problem({**base, "NITER": 10, "NCHUNKS": 2})


def test_fast_result_properties():
    r = np.array([0.5, 0.25, 1.0, 0.125])
    res = fast_amd.FastResult(r, 2e-6)
    np.testing.assert_allclose(res.dB_rel, 10 * np.log10(r))
    np.testing.assert_allclose(res.power, 2e-6 * r)
    np.testing.assert_allclose(res.dBm, 10 * np.log10(r * 2e-6 / 1e-3))
    np.testing.assert_allclose(res.scintillation_index, (r / r.mean()).var())
    np.testing.assert_allclose(res.avg_power_dB_rel, 10 * np.log10(r.mean()))
    assert "Scintillation index" in str(res)


@pytest.mark.parametrize("name", ["temporal_default", "temporal_small", "temporal_noao", "temporal_npxls100"])
def test_temporal_host_setup_matches_reference(name):
    g = load_golden(name)
    p = fast_amd.conf.ConfigParser(dict(params_from_json(g["params_json"]))).config
    prob = host.build_problem(p)
    np.testing.assert_allclose(prob.temporal.pixel_shifts, g["pixel_shifts"], rtol=1e-13)
    np.testing.assert_allclose(prob.temporal.logamp_powerspec, g["temporal_logamp_powerspec"], rtol=1e-9, atol=1e-30)


def test_fits_round_trip(tmp_path):
    from fast_amd import fitsio
    data = np.random.default_rng(0).random(37) * 1e-6
    hdr = {"ZENITH": 55, "WVL": 1550, "OTRSCALE": "inf", "INRSCALE": 1e-6, "AO_MODE": "AO", "ALIAS": "True",
           "W0": 0.3567852394658434, "DIFFLIM": 3.111218108226167e-06, "NITER": 100, "SEED": 1}
    f = str(tmp_path / "r.fits")
    fitsio.writeto(f, data, header=hdr)
    assert len(open(f, "rb").read()) % 2880 == 0
    h2, d2 = fitsio.read(f)
    assert np.array_equal(d2, data)
    for k, v in hdr.items():
        assert h2[k] == v, k
    assert h2["BITPIX"] == -64 and h2["NAXIS1"] == 37
    with pytest.raises(OSError):
        fitsio.writeto(f, data, header=hdr)
    fitsio.writeto(f, data[:5], header=hdr, overwrite=True)
    res = fast_amd.load(f)
    np.testing.assert_allclose(res.power, data[:5], rtol=1e-15)
    np.testing.assert_allclose(res._r, data[:5] / hdr["DIFFLIM"], rtol=1e-15)
    assert res.hdr["SEED"] == 1


def test_optional_rounding_of_auto_grid_size():
    g = load_golden("e2e_default164")
    p = fast_amd.conf.ConfigParser(dict(params_from_json(g["params_json"]))).config
    assert fast_amd.conf.GPU_DEFAULTS["GPU_ROUND_NPXLS"] is False and host.build_problem(dict(p)).N == 164      # default (round 6): the reference's grid
    p["GPU_ROUND_NPXLS"] = False
    assert host.build_problem(dict(p)).N == 164                      # the reference's auto rule
    p["GPU_ROUND_NPXLS"] = True
    assert host.build_problem(dict(p)).N == 192                      # the next multiple of 64: three sub-rows of 64 points, eight rows per wavefront
    p["GPU_ROUND_NPXLS"] = "auto"                                    # (the default of rounds 3-5) round unless the grid is tied to the reference's
    for rng_mode, temporal, n in (("device", False, 192), ("host", False, 164), ("device", True, None)):
        q = dict(p, GPU_RNG=rng_mode, TEMPORAL=temporal)
        if n is not None:
            assert host.build_problem(q).N == n
        else:
            assert host.build_problem(q).N not in (192,) or host.build_problem(dict(q, GPU_ROUND_NPXLS=False)).N == 192
    assert host.WAVE_FFT_SIZES == [128, 192, 256, 320, 384, 448, 512, 576, 640, 768, 896, 1024, 1152, 1280, 1536, 1792, 2048, 4096]
    # rounding goes to the smallest fast grid within 10 % of the best rate at or above N: since round 6 every multiple of 64 is one
    # (packed sub-rows), so it is the next multiple of 64 nearly everywhere (2048 beats 1856 ... 1984 by less than 10 %)
    assert [host.round_up_size(n) for n in (90, 101, 164, 510, 820, 1030, 1290, 1700, 2050, 2310, 3100, 4097)] == \
        [128, 128, 192, 512, 832, 1088, 1344, 1728, 2112, 2368, 3136, None]
    assert all(host.round_up_size(n) - n < 128 for n in range(129, 4096, 7))       # (3072 and 3840, 256-point sub-rows, beat the multiple of 64 below them)
    from oracle import devrng
    for n in host.ROUND_UP_SIZES:       # every listed size has an FFT kernel family in the library
        assert n in host.WAVE_FFT_SIZES or devrng.pks_split(n) or devrng.mr_supported(n), n


def test_comms_host_logic_without_gpu(monkeypatch):
    """fast_amd.comms forms fade_prob / fade_dur / BER from the four numbers per query the device returns;
    with the device reduction replaced by its numpy definition (include/fastmc.h) the host logic alone must
    reproduce the reference's values (comms_metrics.npz), NaN conventions included."""
    from scipy.special import erfc
    from fast_amd import comms, _lib

    def fake_link_metrics(queries, samples=None, handle=None, device=0):
        x = np.asarray(samples, dtype=float)
        out = np.zeros((len(queries), 4))
        for i, (kind, p0, p1) in enumerate(queries):
            if kind == _lib.LM_FADE:
                below = x < p0
                clear = np.flatnonzero(~below)
                out[i] = [below.sum(), np.sum(below[1:] & ~below[:-1]), clear[0] if len(clear) else len(x),
                          clear[-1] if len(clear) else -1]
            else:
                s = x / x.mean()
                if kind == _lib.LM_BER_OOK:
                    v = 0.5 * erfc(s * np.sqrt(10 ** (p0 / 10)) / np.sqrt(2))
                else:
                    q = 0.5 * erfc(np.sqrt(3 / (p0 - 1) * 10 ** (p1 / 10) * s ** 2) / np.sqrt(2))
                    c = (np.sqrt(p0) - 1) / np.sqrt(p0)
                    v = 4 * (c * q - c * c * q * q)
                out[i] = [v.sum(), x.mean(), len(x), 0]
        return out

    monkeypatch.setattr(_lib, "link_metrics", fake_link_metrics)
    monkeypatch.setattr(_lib, "default_device", lambda: 0)
    d = load_golden("comms_metrics")
    thr, eb, Ms, dt = d["thresholds"], d["ebn0"], d["Ms"], float(d["dt"])
    for n in d["names"]:
        v = d["v_" + n]
        np.testing.assert_array_equal([comms.fade_prob(v, t) for t in thr], d["fade_prob_" + n])
        np.testing.assert_array_equal([comms.fade_prob(v, t, 5) for t in thr], d["fade_prob_min5_" + n])
        np.testing.assert_allclose([comms.fade_dur(v, t, dt) for t in thr], d["fade_dur_" + n], rtol=1e-14)
        np.testing.assert_allclose([comms.fade_dur(v, t, dt, 5) for t in thr], d["fade_dur_min5_" + n], rtol=1e-14)
        np.testing.assert_allclose([comms.ber_ook(s, v) for s in eb], d["ber_ook_" + n], rtol=1e-12)
        np.testing.assert_allclose([[comms.ber_qam(M, s, v) for s in eb] for M in Ms], d["ber_qam_" + n], rtol=1e-12)
    np.testing.assert_allclose([comms.ber_ook(s) for s in eb], d["ber_ook_nosamples"], rtol=1e-13)


def test_pupil_cache_entries_are_complete_and_read_only():
    """A failed init must not leave a half-built cache entry (the next Fast() with the same key then raises the
    reference's exception again, fast/funcs.py:298-299, not an AttributeError), and the arrays shared through the cache
    cannot be edited in place."""
    p = params_from_json(load_golden("e2e_axicon")["params_json"])
    p["W0"] = "opt"                                     # axicon + W0 'opt': TypeError in the reference
    for _ in range(2):
        with pytest.raises(TypeError, match="axicon"):
            host.build_problem(fast_amd.conf.ConfigParser(dict(p)).config)
    g, cfg, prob = _problem("ao_alias")
    assert host.build_problem(cfg).pup is prob.pup       # served from the cache
    assert isinstance(prob.pup.token, int) and prob.pup.token > 0
    with pytest.raises(ValueError, match="read-only"):
        prob.pup.pupil[0, 0] = 7.0
    with pytest.raises(ValueError, match="read-only"):
        prob.pup.pupil_filter[0, 0] = 7.0


def test_theta0_is_in_arcseconds_like_aotools():
    """aotools.isoplanaticAngle returns arcseconds; the reference stores it as theta0 / theta0_los and in THETA0."""
    cn2, h = np.array([1e-13, 2e-14]), np.array([1000.0, 9000.0])
    rad = 0.057 * 500e-9 ** 1.2 * np.sum(cn2 * h ** (5 / 3)) ** -0.6
    assert hostmath.isoplanatic_angle(cn2, h) == pytest.approx(rad * 206264.80624709636, rel=1e-12)
    assert 0.5 < hostmath.isoplanatic_angle(cn2, h) < 10        # arcseconds at 500 nm, not 1e-5


def test_grid_beyond_the_kernels_limit_is_a_clear_exception():
    """The reference has no upper limit on NPXLS (fast.py:176-211); the GPU kernels stop at 8192 (beyond 4096: windows of up to
    256 pixels, or the sub-row grids).  A larger grid -- explicit, or
    auto-sized by a long TEMPORAL series (fast.py:201-206: half the total wind displacement) -- fails in the host set-up with an
    Exception that names the limit, before any O(N^2) work and without touching the GPU."""
    import fast_amd
    from conftest import load_golden, params_from_json
    p = params_from_json(load_golden("e2e_ao_alias")["params_json"])
    with pytest.raises(Exception, match="exceeds the GPU kernels' limit of 8192"):
        fast_amd.Fast(dict(p, NPXLS=8194))
    with pytest.raises(Exception, match="exceeds the GPU kernels' limit of 4096.*window of at most 256"):
        fast_amd.Fast(dict(p, NPXLS=4098, D_GROUND=3.0, DX=0.01))          # a 302-pixel pupil window on a grid without sub-rows
    q = params_from_json(load_golden("temporal_default")["params_json"])
    q.update({"NITER": 200000, "NCHUNKS": 10, "DT": 0.01})        # 32.6 m/s x 0.01 s x 200 000 steps / 1 cm / 2 = 3.3e6 columns
    with pytest.raises(Exception, match="limit of 8192.*TEMPORAL"):
        fast_amd.Fast(q)


def test_grids_beyond_4096_that_the_kernels_serve():
    """host.big_grid_supported restates fastmc_create's rule: N = 64 P S or 50 P S, S <= 8 sub-rows of 7 <= P <= 24 values per
    lane (P = 2^k times 1, 3, 5, 7 or 9), up to 8192."""
    from fast_amd import host
    yes = [4608, 4800, 5000, 5120, 6000, 6144, 6400, 7000, 7168, 7680, 8000, 8192]
    no = [4096, 4098, 4100, 5632, 8193, 8256, 9000, 9600, 12288]
    assert all(host.big_grid_supported(n) for n in yes) and not any(host.big_grid_supported(n) for n in no)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import devrng
    for n in yes:      # the restated stream layout knows them too: N / 16 or N / 8 streams on the multiples of 64 (packed sub-rows), else 50 S
        assert devrng.stream_lanes(n) == ((n // 16 if n % 128 == 0 else n // 8) if n % 64 == 0 else 50 * devrng.mr_split(n)) and devrng.stream_lanes(n) > 64
