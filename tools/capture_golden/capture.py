#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference IN THIS CONTAINER.

Runs only where /root/reference exists (the build container).  The reference needs
aotools / astropy / skyfield / pyfftw, none of which is installed; `shims/` holds our
own stand-ins (see shims/README.md).  Nothing of the reference travels: the fixtures
are data -- inputs and the outputs the reference computed for them.

    python tools/capture_golden/capture.py            # rewrites tests/golden/

Fixture families (see tests/golden/MANIFEST.md, written by this script):
  kat_fft_*      reference funcs.make_phase_fft (FFTW branch) on explicit coefficients
  kat_detector   reference Fast.compute_detector on explicit phase cubes
  kat_detector_explicit_W, e2e_explicit_pupil*   the same and whole runs with the pupil weights written out here: no aotools stand-in in the chain
  kat_vk / kat_subharm / kat_simpson   small function-level known answers
  e2e_*          full fast.Fast(config).run(): every init product + result._r
  big_*          1024^2 runs: scalars, strided spectrum sample, first powers
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import fast  # noqa: E402  (the reference)
from fast import funcs, ao_power_spectra, turbulence_models  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden"))
MANIFEST = []


def save(name, note, standin, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    MANIFEST.append((name, os.path.getsize(path), "yes" if standin else "no", note))
    print(f"  {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def params_to_json(p):
    out = {}
    for k, v in p.items():
        if isinstance(v, np.ndarray):
            out[k] = {"__ndarray__": v.tolist()}
        elif isinstance(v, float) and np.isinf(v):
            out[k] = {"__float__": "inf"}
        elif isinstance(v, (np.floating, np.integer)):
            out[k] = v.item()
        else:
            out[k] = v
    return json.dumps(out)


def fftw_objs_for(shape):
    import pyfftw  # the stand-in
    o = {"IN": pyfftw.empty_aligned(shape, dtype="complex128"),
         "OUT": pyfftw.empty_aligned(shape, dtype="complex128")}
    o["FFT"] = pyfftw.FFTW(o["IN"], o["OUT"], axes=(-1, -2))
    return o


# ------------------------------------------------------------------ KATs
def kat_fft():
    rng = np.random.default_rng(20240601)
    for N, L0 in ((16, np.inf), (30, 25.0), (33, np.inf), (64, 25.0), (128, np.inf)):
        B = 2
        dx = 0.02
        freq = fast.fast.SpatialFrequencies(N, dx)
        ps = funcs.turb_powerspectrum_vonKarman(freq.main, np.array([3e-13, 1e-13]), L0, 1e-3).sum(0)
        ps = ps * 2 * np.pi * (2 * np.pi / 1550e-9) ** 2
        coeffs = rng.normal(size=(B, N, N)) + 1j * rng.normal(size=(B, N, N))
        rand = coeffs * np.sqrt(ps)
        objs = fftw_objs_for((B, N, N))
        scr = funcs.make_phase_fft(rand, freq.main.df, True, objs, double=True)
        save(f"kat_fft_N{N}", f"make_phase_fft FFTW branch, N={N}, L0={L0}", False,
             N=N, dx=dx, df=freq.main.df, powerspec=ps, coeffs=coeffs, screens=scr)


def base_params(**over):
    h = np.array([500.0, 4000.0, 9000.0, 15000.0])
    cn2 = np.array([4e-13, 9e-14, 4e-14, 1e-14])
    w = np.array([8.0, 15.0, 30.0, 12.0])
    p = dict(fast.conf.DEFAULTS)
    p.update({
        "NPXLS": 64, "DX": 0.01, "NITER": 8, "NCHUNKS": 2, "TEMPORAL": False, "FFTW": True,
        "SEED": 7, "LOGLEVEL": "ERROR", "W0": "opt", "D_GROUND": 0.2, "OBSC_GROUND": 0,
        "D_SAT": 0.1, "H_SAT": 600e3, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w,
        "WIND_DIR": np.array([0.0, 90.0, 180.0, 270.0]), "L0": np.inf, "l0": 1e-6,
        "ZENITH_ANGLE": 30, "DTHETA": [4, 0], "AO_MODE": "AO", "DSUBAP": 0.05,
        "TLOOP": 1e-3, "TEXP": 1e-3, "ALIAS": True, "NOISE": 0.0,
    })
    p.update(over)
    return p


def kat_detector():
    sim = fast.Fast(base_params())
    rng = np.random.default_rng(99)
    M, Np = sim.Niter_per_chunk, sim.Npxls_pup
    phs = rng.normal(scale=6.0, size=(M, Np, Np))
    sim.phs[:] = phs
    sim.logamp[:] = rng.normal(scale=0.2, size=sim.Niter)
    inc = sim.compute_detector(chunk=1).copy()
    sim.params["COHERENT"] = True
    coh = sim.compute_detector(chunk=1).copy()
    save("kat_detector", "Fast.compute_detector on an explicit phase cube (chunk=1)", True,
         phs=phs, W=sim.pupil * sim.pupil_mode, dx=sim.dx, logamp=sim.logamp.copy(), chunk=1,
         M=M, incoherent=inc, coherent=coh)


def explicit_weights(Np, dx, D):
    """Pupil weights written out in numpy (no aotools stand-in in their making): a disc of diameter D sampled at pixel centres
    (x_i = (i - (Np - 1) / 2) dx) times a Gaussian mode of 1/e^2 half-width 0.45 D, both on the Np x Np pupil grid."""
    x = (np.arange(Np) - (Np - 1) / 2.0) * dx
    r2 = x[:, None] ** 2 + x[None, :] ** 2
    disc = (r2 <= (D / 2.0) ** 2).astype(float)
    mode = np.exp(-r2 / (0.45 * D) ** 2)
    return disc, mode


def standin_free():
    """Fixtures for SURVEY 8 rows a5 / a10 with NO stand-in in the chain.  The reference object is built as
    usual (its __init__ runs the aotools stand-ins), then everything the Monte-Carlo path reads that came from a stand-in is
    replaced by arrays written out here -- pupil, fibre mode, pupil_filter = 1 -- and the products that depend on them are
    recomputed by the reference's own code (compute_powerspec: zonal AO + alias, scipy and numpy only).  What is left of the
    stand-ins is pyfftw's: FFTW.__call__ as numpy.fft.fft2 over the planned axes, i.e. the unnormalised forward DFT FFTW computes."""
    # 1. the detector on an explicit phase cube with explicit weights
    sim = fast.Fast(base_params())
    Np = sim.Npxls_pup
    disc, mode = explicit_weights(Np, sim.dx, 0.2)
    sim.pupil, sim.pupil_mode = disc, mode
    rng = np.random.default_rng(123)
    M = sim.Niter_per_chunk
    phs = rng.normal(scale=5.0, size=(M, Np, Np))
    sim.phs[:] = phs
    sim.logamp[:] = rng.normal(scale=0.15, size=sim.Niter)
    inc = sim.compute_detector(chunk=0).copy()
    sim.params["COHERENT"] = True
    coh = sim.compute_detector(chunk=0).copy()
    save("kat_detector_explicit_W", "Fast.compute_detector on an explicit phase cube with weights written out in numpy (chunk=0)", False,
         phs=phs, W=disc * mode, dx=sim.dx, logamp=sim.logamp.copy(), chunk=0, M=M, incoherent=inc, coherent=coh)
    # 2. end to end: AO zonal + alias, 40 iterations, explicit weights, pupil_filter = 1
    for name, over, note in (("e2e_explicit_pupil", dict(NITER=40, NCHUNKS=4, SEED=21), "AO zonal + ALIAS"),
                             ("e2e_explicit_pupil_noao", dict(NITER=40, NCHUNKS=4, SEED=22, AO_MODE="NOAO", L0=30.0, COHERENT=True),
                              "NOAO, L0 = 30 m, COHERENT")):
        p = base_params(**over)
        sim = fast.Fast(p)
        disc, mode = explicit_weights(sim.Npxls_pup, sim.dx, p["D_GROUND"])
        sim.pupil, sim.pupil_mode = disc, mode
        sim.pupil_filter = np.ones((sim.Npxls, sim.Npxls))
        sim.compute_powerspec()
        res = sim.run()
        save(name, note + ": Fast.run() with pupil, fibre mode and pupil_filter replaced by explicit arrays after __init__ and "
             "compute_powerspec() run again -- no aotools stand-in between the config and result._r", False,
             params_json=np.array(params_to_json(p)), W=disc * mode, dx=np.array(sim.dx), df=np.array(sim.freq.df), Npxls=np.array(sim.Npxls),
             Npxls_pup=np.array(sim.Npxls_pup), h=sim.h, cn2=sim.cn2, wind_vector=sim.wind_vector,
             lf_mask=np.asarray(sim.lf_mask, dtype=float), powerspec=sim.powerspec, powerspec_per_layer=sim.powerspec_per_layer,
             logamp_powerspec=sim.logamp_powerspec, logamp_var=np.array(sim.logamp_var), phs_var=np.array(sim.phs_var),
             fitting_error=np.array(sim.fitting_error), aniso_servo_error=np.array(sim.aniso_servo_error),
             alias_error=np.array(sim.alias_error), noise_error=np.array(sim.noise_error),
             r=res._r, logamp=sim.logamp.copy(), phs_last_chunk=sim.phs.copy())


def kat_small():
    # von Karman, stacked and plain
    freq = fast.fast.SpatialFrequencies(24, 0.05)
    cn2 = np.array([2e-13, 5e-14, 1e-14])
    save("kat_vk", "funcs.turb_powerspectrum_vonKarman", False,
         N=24, dx=0.05, cn2=cn2,
         vk_inf=funcs.turb_powerspectrum_vonKarman(freq.main, cn2, np.inf, 1e-6),
         vk_L0=funcs.turb_powerspectrum_vonKarman(freq.main, cn2, 25.0, 0.01),
         simpson_vk_L0=funcs.integrate_powerspectrum(
             funcs.turb_powerspectrum_vonKarman(freq.main, cn2, 25.0, 0.01), freq.main.f))
    # sub-harmonic screens
    N, dx = 20, 0.03
    freq = fast.fast.SpatialFrequencies(N, dx)
    freq.make_subharm_freqs()
    rng = np.random.default_rng(5)
    rand = rng.normal(size=(2, 3, 3, 3)) + 1j * rng.normal(size=(2, 3, 3, 3))
    save("kat_subharm", "funcs.make_phase_subharm(double=True) + subharm grids", False,
         N=N, dx=dx, rand=rand, fx=freq.subharm.fx, fy=freq.subharm.fy, df=freq.subharm.df,
         screens=funcs.make_phase_subharm(rand, freq, N, dx, double=True))
    # turbulence models
    h4, c4, w4 = turbulence_models.HV57_Bufton_profile(4)
    h10, c10, w10 = turbulence_models.HV57_Bufton_profile(10, w=30, A=3e-14, vg=5)
    hh = np.linspace(0, 20000, 10)
    save("kat_turbulence", "turbulence_models.HV57 / Bufton_wind / HV57_Bufton_profile", False,
         h4=h4, cn2_4=c4, w4=w4, h10=h10, cn2_10=c10, w10=w10, hh=hh,
         hv57=turbulence_models.HV57(hh), bufton=turbulence_models.Bufton_wind(hh),
         rng_anchor=np.random.default_rng(1).normal(0, 1, 3),
         l_path=np.array([funcs.l_path(36e6, 55.0), funcs.l_path(600e3, 0.0), funcs.l_path(600e3, 70.0)]),
         wind_corr=funcs.calculate_wind_correction(np.array([1e3, 1e4]), [30.0, -12.0], 1e-3))
    # zernike filters / masks
    freq = fast.fast.SpatialFrequencies(32, 0.02)
    m = freq.main
    save("kat_masks", "ao_power_spectra.mask_lf variants + zernike_squared_filter", True,
         N=32, dx=0.02,
         zonal=ao_power_spectra.mask_lf(m, 0.08),
         modal=ao_power_spectra.mask_lf(m, 0.08, modal=True, modal_mult=0.7),
         zern3=ao_power_spectra.mask_lf(m, 0.08, modal=True, modal_mult=1, Zmax=3, D=0.4),
         zern9=ao_power_spectra.mask_lf(m, 0.08, modal=True, modal_mult=1, Zmax=9, D=0.4),
         zsq4=ao_power_spectra.zernike_squared_filter(m.fabs, m.fx, m.fy, 0.4, 4).real)


# ------------------------------------------------------------------ end-to-end
E2E_KEYS_SCALAR = ["dx", "Npxls", "Npxls_pup", "L", "paa", "r0", "theta0", "tau0", "r0_los",
                   "theta0_los", "tau0_los", "W0", "W0_sat", "diffraction_limit", "logamp_var",
                   "phs_var", "fitting_error", "aniso_servo_error", "alias_error", "noise_error",
                   "zenith_correction", "k", "dx_sat"]


def capture_sim(name, p, note, full=True, stride=None):
    t0 = time.time()
    sim = fast.Fast(p)
    t1 = time.time()
    res = sim.run()
    t2 = time.time()
    d = {"params_json": np.array(params_to_json(p))}
    for k in E2E_KEYS_SCALAR:
        d[k] = np.array(getattr(sim, k))
    d["h"] = sim.h
    d["cn2"] = sim.cn2
    d["wind_vector"] = sim.wind_vector
    d["wind_speed"] = sim.wind_speed
    d["pup_coords"] = sim.pup_coords
    d["link_budget_keys"] = np.array(list(sim.link_budget.keys()))
    d["link_budget_vals"] = np.array(list(sim.link_budget.values()), dtype=float)
    d["phs_var_weights"] = np.asarray(sim.phs_var_weights)
    d["r"] = res._r
    d["logamp"] = sim.logamp.copy()
    d["ref_init_s"] = np.array(t1 - t0)
    d["ref_run_s"] = np.array(t2 - t1)
    if full:
        d["pupil"] = sim.pupil
        d["pupil_mode"] = sim.pupil_mode
        d["pupil_sat"] = sim.pupil_sat
        d["pupil_mode_sat"] = sim.pupil_mode_sat
        d["pupil_filter"] = sim.pupil_filter
        d["lf_mask"] = np.asarray(sim.lf_mask)
        d["powerspec"] = sim.powerspec
        d["powerspec_per_layer"] = sim.powerspec_per_layer
        d["turb_powerspec"] = sim.turb_powerspec
        d["G_ao"] = np.asarray(sim.G_ao, dtype=float)
        d["alias_powerspec"] = np.asarray(sim.alias_powerspec, dtype=float)
        d["noise_powerspec"] = np.asarray(sim.noise_powerspec, dtype=float)
        d["logamp_powerspec"] = sim.logamp_powerspec
        d["phs_last_chunk"] = sim.phs.copy()
        if sim.subharmonics:
            d["powerspec_subharm"] = sim.powerspec_subharm
            d["sh_fx"] = sim.freq.subharm.fx
            d["sh_fy"] = sim.freq.subharm.fy
            d["sh_df"] = sim.freq.subharm.df
            # the sub-harmonic bookkeeping a caller can read off the object (fast.py:494-526)
            d["powerspec_subharm_per_layer"] = sim.powerspec_subharm_per_layer
            d["phs_var_subharm"] = sim.phs_var_subharm
            d["phs_var_weights_sh"] = sim.phs_var_weights_sh
            d["lf_mask_subharm"] = np.asarray(sim.lf_mask_subharm, dtype=float)
    else:
        s = stride
        d["stride"] = np.array(s)
        d["powerspec_strided"] = sim.powerspec[::s, ::s].copy()
        d["powerspec_centre"] = sim.powerspec[sim.Npxls // 2 - 16: sim.Npxls // 2 + 16,
                                              sim.Npxls // 2 - 16: sim.Npxls // 2 + 16].copy()
        d["powerspec_sum"] = np.array(sim.powerspec.sum())
        d["W"] = sim.pupil * sim.pupil_mode
        d["lf_mask_sum"] = np.array(np.asarray(sim.lf_mask).sum())
    save(name, note + f" [ref init {t1-t0:.2f}s run {t2-t1:.2f}s]", True, **d)


def e2e(only=None):
    cases = {
        "ao_alias": (dict(), "AO zonal + ALIAS"),
        "noao": (dict(AO_MODE="NOAO"), "NOAO, L0=inf"),
        "noao_L0": (dict(AO_MODE="NOAO", L0=25.0, l0=0.005), "NOAO, L0=25, l0=5mm"),
        "tt": (dict(AO_MODE="TT"), "TT (modal, Zmax=3)"),
        "noise": (dict(NOISE=0.3), "AO + ALIAS + NOISE 0.3"),
        "noalias_noise": (dict(ALIAS=False, NOISE=1.0), "AO, no alias, NOISE 1"),
        "modal": (dict(MODAL=True, MODAL_MULT=0.8), "AO modal radial cut"),
        "modal_zmax": (dict(MODAL=True, ZMAX=10), "AO modal Zernike Zmax=10"),
        "lgsao": (dict(AO_MODE="LGSAO"), "LGSAO"),
        "subharm": (dict(SUBHARM=True, AO_MODE="NOAO", L0=40.0), "NOAO + SUBHARM, L0=40"),
        "subharm_ao": (dict(SUBHARM=True), "AO + ALIAS + SUBHARM"),
        "coherent": (dict(COHERENT=True), "AO, COHERENT detection"),
        "down": (dict(PROP_DIR="down"), "downlink"),
        "obsc": (dict(OBSC_GROUND=0.05, OBSC_SAT=0.03), "central obscurations"),
        "axicon": (dict(W0=0.06, AXICON=True, OBSC_GROUND=0.05), "axicon launch, fixed W0"),
        "w0fixed": (dict(W0=0.07), "fixed W0"),
        "lsat_aniso": (dict(L_SAT=500e3, ANISO_DL=[20.0, -10.0], AZIMUT_SAT=35.0, DTHETA=[3.0, 2.0]),
                       "L_SAT + ANISO_DL + AZIMUT_SAT optional keys"),
        "oddNp": (dict(D_GROUND=0.205, NPXLS=48), "Np odd (23), N=48 (non power of two)"),
        "autosize": (dict(NPXLS="auto", DX="auto", NITER=4, NCHUNKS=1), "auto DX and NPXLS"),
        "oddN": (dict(NPXLS=49, SUBHARM=True), "odd grid size N=49 (numpy's asymmetric fftshift), sub-harmonics"),
    }
    for name, (over, note) in cases.items():
        if only and name not in only:
            continue
        capture_sim("e2e_" + name, base_params(**over), note)


def temporal():
    """TEMPORAL (frozen-flow) fixtures: the shipped test_params.py geometry and a small one."""
    def cap(name, p, note):
        sim = fast.Fast(p)
        res = sim.run()
        d = {"params_json": np.array(params_to_json(p))}
        for k in ("dx", "Npxls", "Npxls_pup", "logamp_var", "diffraction_limit", "W0"):
            d[k] = np.array(getattr(sim, k))
        d.update(h=sim.h, cn2=sim.cn2, wind_vector=sim.wind_vector, wind_speed=sim.wind_speed,
                 wind_dir=np.asarray(sim.wind_dir, dtype=float), pupil=sim.pupil, pupil_mode=sim.pupil_mode,
                 powerspec_per_layer=sim.powerspec_per_layer, temporal_logamp_powerspec=sim.temporal_logamp_powerspec,
                 pixel_shifts=sim.pixel_shifts, logamp=sim.logamp.copy(), r=res._r, phs_last_chunk=sim.phs.copy(),
                 fx_axis_t=sim.freq.temporal.fx_axis, fy_axis_t=sim.freq.temporal.fy_axis, fabs_t=sim.freq.temporal.fabs)
        save(name, note, True, **d)
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    p = dict(fast.conf.DEFAULTS)
    p.update({"NPXLS": "auto", "DX": 0.01, "NITER": 100, "NCHUNKS": 10, "TEMPORAL": True, "FFTW": True, "SEED": 1,
              "DT": 0.001, "W0": "opt", "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w,
              "WIND_DIR": [0, 90, 180, 270], "ZENITH_ANGLE": 55, "DSUBAP": 0.1, "LOGLEVEL": "ERROR", "H_SAT": 36e6})
    cap("temporal_default", p, "test/test_params.py as shipped (TEMPORAL True) with FFTW True, SEED 1")
    cap("temporal_small", base_params(TEMPORAL=True, NITER=24, NCHUNKS=4, DT=0.004, NPXLS=64),
        "small TEMPORAL config: N=64, 24 steps of 4 ms, wrap-around of the sample coordinates")
    cap("temporal_noao", base_params(TEMPORAL=True, NITER=12, NCHUNKS=3, DT=0.01, NPXLS=48, AO_MODE="NOAO", L0=20.0, COHERENT=True),
        "TEMPORAL + NOAO + COHERENT, N=48")


def mean_irradiance():
    sim = fast.Fast(base_params())
    sim2 = fast.Fast(base_params(AO_MODE="NOAO", L0=30.0, NPXLS=48))
    save("mean_irradiance", "Fast.compute_mean_irradiance on- and off-axis (aotools ft2/ift2 stand-ins)", True,
         powerspec=sim.powerspec, W=sim.pupil * sim.pupil_mode, dx=sim.dx, df=sim.freq.df,
         diffraction_limit=sim.diffraction_limit, onaxis=sim.compute_mean_irradiance(),
         offaxis=sim.compute_mean_irradiance(onaxis=False),
         powerspec2=sim2.powerspec, W2=sim2.pupil * sim2.pupil_mode, dx2=sim2.dx, df2=sim2.freq.df,
         diffraction_limit2=sim2.diffraction_limit, onaxis2=sim2.compute_mean_irradiance())


def stat_ref():
    """2000 iterations of the reference at 256^2 (BASELINE config 1 geometry, AO + alias) and 2000 without
    AO: only the result vectors, for distribution tests of the device-generator path."""
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    p = dict(fast.conf.DEFAULTS)
    p.update({"NPXLS": 256, "DX": 0.01, "NITER": 2000, "NCHUNKS": 20, "TEMPORAL": False, "FFTW": True, "SEED": 11,
              "W0": "opt", "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": [0, 90, 180, 270],
              "ZENITH_ANGLE": 55, "DSUBAP": 0.1, "LOGLEVEL": "ERROR", "H_SAT": 36e6, "AO_MODE": "AO", "ALIAS": True})
    sim = fast.Fast(p)
    r_ao = sim.run()._r
    p2 = dict(p)
    p2.update({"AO_MODE": "NOAO", "L0": 25.0, "SEED": 12})
    sim2 = fast.Fast(p2)
    r_no = sim2.run()._r
    save("stat_ref_256", "result._r of 2000 reference iterations at 256^2: AO+alias, and NOAO L0=25", True,
         params_json=np.array(params_to_json(p)), params2_json=np.array(params_to_json(p2)), r_ao=r_ao, r_noao=r_no)


def stat_ref_256b():
    """8000 more iterations of each stat_ref configuration (other seeds): with stat_ref_256 a reference sample of 10 000 per
    configuration, so that the 256^2 distribution test can hold the scintillation index to 10 % (bootstrap s.e. of 2000: 3.9 %)."""
    g = np.load(os.path.join(OUT, "stat_ref_256.npz"))
    out = {}
    for key, pkey, seed in (("r_ao", "params_json", 111), ("r_noao", "params2_json", 112)):
        raw = json.loads(str(g[pkey]))
        p = dict(fast.conf.DEFAULTS)
        for k, v in raw.items():
            p[k] = np.array(v["__ndarray__"]) if isinstance(v, dict) and "__ndarray__" in v else (
                float(v["__float__"]) if isinstance(v, dict) and "__float__" in v else v)
        p.update({"NITER": 8000, "NCHUNKS": 80, "SEED": seed})
        out[key] = fast.Fast(p).run()._r
    save("stat_ref_256b", "8000 more reference iterations of each stat_ref_256 configuration (SEED 111 / 112)", True, **out)


def stat_ref_1024():
    """4000 iterations of the reference at the BENCHMARKED size (BASELINE configs[1]: 1024^2, NOAO, L0 = 25 m -- the hard
    26 rad case -- and configs[2]: AO + alias): only the result vectors, for distribution tests of the default (float64
    device generator) path at the size the bench line is quoted on.  ~5 min each at 13.8 it/s."""
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    p = dict(fast.conf.DEFAULTS)
    p.update({"NPXLS": 1024, "DX": 0.01, "NITER": 4000, "NCHUNKS": 400, "TEMPORAL": False, "FFTW": True, "SEED": 21,
              "W0": "opt", "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": [0, 90, 180, 270],
              "ZENITH_ANGLE": 55, "DSUBAP": 0.1, "LOGLEVEL": "ERROR", "H_SAT": 36e6, "AO_MODE": "NOAO", "L0": 25.0})
    t0 = time.time()
    r_no = fast.Fast(p).run()._r
    t1 = time.time()
    p2 = dict(p)
    p2.update({"AO_MODE": "AO", "ALIAS": True, "L0": np.inf, "SEED": 22, "NOISE": 0.0, "DTHETA": [4, 0], "TLOOP": 1e-3, "TEXP": 1e-3})
    r_ao = fast.Fast(p2).run()._r
    t2 = time.time()
    print(f"  reference: NOAO {t1 - t0:.0f} s, AO {t2 - t1:.0f} s")
    save("stat_ref_1024", "result._r of 4000 reference iterations at 1024^2: configs[1] NOAO L0=25, and configs[2] AO+alias", True,
         params_json=np.array(params_to_json(p)), params2_json=np.array(params_to_json(p2)), r_noao=r_no, r_ao=r_ao)


def comms_metrics():
    """Reference fade statistics and BER / SEP integrals (comms.py:171-262) on explicit sample vectors:
    a correlated log-normal series with many fades, its edge-case slices, and the stat_ref NOAO powers."""
    from fast import comms
    rng = np.random.default_rng(2024)
    n = 40000
    e = rng.normal(0, 1, n)
    x = np.empty(n)
    a = 0.97
    x[0] = e[0]
    for i in range(1, n):
        x[i] = a * x[i - 1] + np.sqrt(1 - a * a) * e[i]
    series = np.exp(0.6 * x - 0.18)                        # mean ~1, fades below 0.3 last tens of samples
    d = np.load(os.path.join(OUT, "stat_ref_256.npz"))
    vectors = {"series": series, "noao": d["r_noao"], "ao": d["r_ao"],
               "starts_in_fade": series[np.argmax(series < 0.3):][:6000],
               "ends_in_fade": series[:len(series) - np.argmax(series[::-1] < 0.3)][-6000:],
               "all_below": np.full(50, 0.1), "none_below": np.full(50, 2.0), "few": series[:400]}
    thresholds = np.array([0.1, 0.3, 0.5, 1.0])
    ebn0 = np.array([0.0, 6.0, 10.0, 14.0])
    Ms = np.array([4, 16, 64])
    out = {}
    for name, v in vectors.items():
        out["v_" + name] = v
        out["fade_prob_" + name] = np.array([comms.fade_prob(v, t) for t in thresholds])
        out["fade_prob_min5_" + name] = np.array([comms.fade_prob(v, t, min_fades=5) for t in thresholds])
        out["fade_dur_" + name] = np.array([comms.fade_dur(v, t, dt=2e-3) for t in thresholds])
        out["fade_dur_min5_" + name] = np.array([comms.fade_dur(v, t, dt=2e-3, min_fades=5) for t in thresholds])
        out["ber_ook_" + name] = np.array([comms.ber_ook(s, v) for s in ebn0])
        out["sep_qam_" + name] = np.array([[comms.sep_qam(M, s, v) for s in ebn0] for M in Ms])
        out["ber_qam_" + name] = np.array([[comms.ber_qam(M, s, v) for s in ebn0] for M in Ms])
    out["ber_ook_nosamples"] = np.array([comms.ber_ook(s) for s in ebn0])
    out["sep_qam_nosamples"] = np.array([[comms.sep_qam(M, s) for s in ebn0] for M in Ms])
    out["ber_qam_nosamples"] = np.array([[comms.ber_qam(M, s) for s in ebn0] for M in Ms])
    save("comms_metrics", "reference comms.fade_prob / fade_dur / ber_ook / sep_qam / ber_qam on explicit sample vectors",
         False, names=np.array(list(vectors)), thresholds=thresholds, ebn0=ebn0, Ms=Ms, dt=np.array(2e-3), **out)


def default_cfg():
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    p = dict(fast.conf.DEFAULTS)
    p.update({"NPXLS": "auto", "DX": 0.01, "NITER": 20, "NCHUNKS": 2, "TEMPORAL": False, "FFTW": True,
              "SEED": 1, "W0": "opt", "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w,
              "WIND_DIR": [0, 90, 180, 270], "ZENITH_ANGLE": 55, "DSUBAP": 0.1, "LOGLEVEL": "ERROR",
              "H_SAT": 36e6, "AO_MODE": "AO", "ALIAS": True})
    capture_sim("e2e_default164", p, "test/test_params.py with TEMPORAL False, FFTW True, SEED 1 (auto N=164)")
    p256 = dict(p)
    p256.update({"NPXLS": 256, "NITER": 100, "NCHUNKS": 10})
    capture_sim("cfg1_256", p256, "BASELINE config 1: NPXLS 256, NITER 100, NCHUNKS 10", full=False, stride=8)
    return p


def big(p):
    pb = dict(p)
    pb.update({"NPXLS": 1024, "NITER": 8, "NCHUNKS": 2, "AO_MODE": "NOAO", "SEED": 3})
    capture_sim("big_noao_1024", pb, "BASELINE config 2 geometry, NITER 8", full=False, stride=16)
    pb2 = dict(pb)
    pb2.update({"L0": 25.0})
    capture_sim("big_noao_L0_1024", pb2, "config 2 with L0=25", full=False, stride=16)
    pb3 = dict(pb)
    pb3.update({"AO_MODE": "AO", "ALIAS": True})
    capture_sim("big_ao_1024", pb3, "BASELINE config 3 geometry (AO + ALIAS), NITER 8", full=False, stride=16)


def big_seeds():
    """SURVEY 8(d) config 2 asks for seeds 1..3 with L0 = inf and L0 = 25: `big()` holds SEED 3, these are SEED 1 and 2."""
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    p = dict(fast.conf.DEFAULTS)
    p.update({"NPXLS": 1024, "DX": 0.01, "NITER": 8, "NCHUNKS": 2, "TEMPORAL": False, "FFTW": True, "W0": "opt",
              "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": [0, 90, 180, 270], "ZENITH_ANGLE": 55,
              "DSUBAP": 0.1, "LOGLEVEL": "ERROR", "H_SAT": 36e6, "AO_MODE": "NOAO", "ALIAS": True})
    for seed in (1, 2):
        for tag, L0 in (("noao", np.inf), ("noao_L0", 25.0)):
            q = dict(p)
            q.update({"SEED": seed, "L0": L0})
            capture_sim(f"big_{tag}_1024_s{seed}", q, f"BASELINE config 2 geometry, L0={L0}, SEED {seed}, NITER 8", full=False, stride=16)


def big2048():
    """BASELINE configs[3] geometry (2048^2), 4 iterations of the reference: the grid the wave kernels
    transform as two interleaved sub-rows."""
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    p = dict(fast.conf.DEFAULTS)
    p.update({"NPXLS": 2048, "DX": 0.01, "NITER": 4, "NCHUNKS": 2, "TEMPORAL": False, "FFTW": True, "SEED": 5, "W0": "opt",
              "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": [0, 90, 180, 270], "ZENITH_ANGLE": 55,
              "DSUBAP": 0.1, "LOGLEVEL": "ERROR", "H_SAT": 36e6, "AO_MODE": "NOAO", "L0": 25.0})
    capture_sim("big_noao_L0_2048", p, "BASELINE configs[3] geometry (2048^2), NOAO, L0=25, NITER 4", full=False, stride=32)
    if "--with-4096" in sys.argv:
        p4 = dict(p)
        p4.update({"NPXLS": 4096, "NITER": 2, "NCHUNKS": 1, "SEED": 6})
        capture_sim("big_noao_L0_4096", p4, "4096^2 (four interleaved sub-rows on the GPU), NOAO, L0=25, NITER 2", full=False, stride=64)


def big_modes():
    """1024^2 runs of the AO modes whose masks use the Zernike / Bessel filters (TT, LGSAO, modal Zmax) and of the
    noise term: 4 iterations each, scalars + strided spectrum sample + powers."""
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    base = dict(fast.conf.DEFAULTS)
    base.update({"NPXLS": 1024, "DX": 0.01, "NITER": 4, "NCHUNKS": 2, "TEMPORAL": False, "FFTW": True, "SEED": 9, "W0": "opt",
                 "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": [0, 90, 180, 270], "ZENITH_ANGLE": 55,
                 "DSUBAP": 0.1, "LOGLEVEL": "ERROR", "H_SAT": 36e6, "ALIAS": True})
    for name, over, note in (("big_tt_1024", {"AO_MODE": "TT"}, "TT (Zernike Zmax=3 mask)"),
                             ("big_lgsao_1024", {"AO_MODE": "LGSAO"}, "LGSAO (Z<=4 filter)"),
                             ("big_modal_zmax_noise_1024", {"AO_MODE": "AO", "MODAL": True, "ZMAX": 21, "NOISE": 0.5},
                              "AO modal Zmax=21 + alias + noise")):
        p = dict(base)
        p.update(over)
        if "--only-combo" not in sys.argv:
            capture_sim(name, p, f"1024^2 {note}, NITER 4", full=False, stride=16)
    p = dict(base)
    p.update({"AO_MODE": "AO", "SUBHARM": True, "COHERENT": True, "PROP_DIR": "down", "L0": 40.0, "SEED": 10})
    capture_sim("big_subharm_coherent_down_1024", p, "1024^2 AO + alias + SUBHARM + COHERENT + downlink, L0=40, NITER 4", full=False, stride=16)


def zenith():
    """BASELINE configs[4]: the zenith-angle scan of fast_amd.sweep.zenith_scan / bench.py (32 angles, linspace(0, 70, 32),
    1024^2, AO + alias, SEED + index) at two of its angles (indices 5 and 27), 8 iterations of the reference each:
    scalars, strided spectrum sample, powers."""
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    angles = np.linspace(0, 70, 32)
    for idx in (5, 27):
        p = dict(fast.conf.DEFAULTS)
        p.update({"NPXLS": 1024, "DX": 0.01, "NITER": 8, "NCHUNKS": 1, "TEMPORAL": False, "SUBHARM": False, "FFTW": True,
                  "SEED": 1 + idx, "W0": "opt", "D_GROUND": 0.8, "OBSC_GROUND": 0, "D_SAT": 0.1, "H_SAT": 36e6, "H_TURB": h,
                  "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": np.array([0., 90., 180., 270.]), "L0": np.inf, "l0": 1e-6,
                  "ZENITH_ANGLE": float(angles[idx]), "DTHETA": [4, 0], "AO_MODE": "AO", "DSUBAP": 0.1, "TLOOP": 1e-3,
                  "TEXP": 1e-3, "ALIAS": True, "NOISE": 0, "LOGLEVEL": "ERROR"})
        capture_sim(f"big_zenith{idx:02d}_1024", p, f"BASELINE configs[4]: zenith scan sample {idx} of 32 ({angles[idx]:.2f} deg), "
                    "1024^2 AO + alias, NITER 8", full=False, stride=16)


def numpy_branch():
    """The reference's DEFAULT transform branch (FFTW False, fast/conf.py:71 -> funcs.py:216-218:
    aotools.fouriertransform.ift2(rand * df, 1.)) on the small AO + alias geometry, same SEED as e2e_ao_alias.
    STAND-IN DEPENDENT in its very arithmetic: ift2 is our stand-in of aotools (N = DATA.shape[0], all-axes
    ifftshift), which on the (chunk, N, N) arrays of Fast.compute_phs scales the screens by (chunk / N)^2 and
    rolls the chunk axis.  Kept to document what `FFTW: False` means next to the FFTW branch the GPU computes."""
    p = base_params(FFTW=False, NITER=20, NCHUNKS=2)
    sim = fast.Fast(p)
    res = sim.run()
    p2 = base_params(FFTW=True, NITER=20, NCHUNKS=2)
    sim2 = fast.Fast(p2)
    res2 = sim2.run()
    save("e2e_numpy_branch", "FFTW False (reference default): aotools ift2 stand-in branch next to the FFTW branch, same SEED, "
         "N=64, 2 chunks of 10", True, params_json=np.array(params_to_json(p)), r=res._r, phs_last_chunk=sim.phs.copy(),
         r_fftw=res2._r, phs_last_chunk_fftw=sim2.phs.copy(), logamp=sim.logamp.copy(), powerspec=sim.powerspec,
         W=sim.pupil * sim.pupil_mode, dx=np.array(sim.dx), df=np.array(sim.freq.main.df), logamp_var=np.array(sim.logamp_var),
         Npxls=np.array(sim.Npxls), Npxls_pup=np.array(sim.Npxls_pup))


def decimal():
    """Round decimal grids (NPXLS 100, 150, 200, 1000: the 50-lane kernel family of the GPU library, 50 streams per row):
    make_phase_fft KATs at N = 100 (N = 0 mod 4) and 150 (N = 2 mod 4), three small end-to-end runs, a TEMPORAL run and
    the BASELINE geometry at 1000^2 (scalars + strided spectrum sample + powers)."""
    rng = np.random.default_rng(20261003)
    for N, L0 in ((100, 25.0), (150, np.inf)):
        B, dx = 2, 0.02
        freq = fast.fast.SpatialFrequencies(N, dx)
        ps = funcs.turb_powerspectrum_vonKarman(freq.main, np.array([3e-13, 1e-13]), L0, 1e-3).sum(0)
        ps = ps * 2 * np.pi * (2 * np.pi / 1550e-9) ** 2
        coeffs = rng.normal(size=(B, N, N)) + 1j * rng.normal(size=(B, N, N))
        scr = funcs.make_phase_fft(coeffs * np.sqrt(ps), freq.main.df, True, fftw_objs_for((B, N, N)), double=True)
        save(f"kat_fft_N{N}", f"make_phase_fft FFTW branch, N={N}, L0={L0}", False,
             N=N, dx=dx, df=freq.main.df, powerspec=ps, coeffs=coeffs, screens=scr)
    capture_sim("e2e_npxls100", base_params(NPXLS=100), "AO + ALIAS on a 100^2 grid")
    capture_sim("e2e_npxls150", base_params(NPXLS=150, SUBHARM=True, AO_MODE="NOAO", L0=40.0, D_GROUND=0.3),
                "NOAO + SUBHARM on a 150^2 grid (N = 2 mod 4), L0=40, 30 cm aperture")
    capture_sim("e2e_npxls200", base_params(NPXLS=200, COHERENT=True, PROP_DIR="down", D_GROUND=0.5, NITER=4, NCHUNKS=2),
                "AO + ALIAS, COHERENT, downlink on a 200^2 grid, 50 cm aperture")
    # the temporal fixture keeps what temporal() keeps
    def cap_t(name, p, note):
        sim = fast.Fast(p)
        res = sim.run()
        d = {"params_json": np.array(params_to_json(p))}
        for k in ("dx", "Npxls", "Npxls_pup", "logamp_var", "diffraction_limit", "W0"):
            d[k] = np.array(getattr(sim, k))
        d.update(h=sim.h, cn2=sim.cn2, wind_vector=sim.wind_vector, wind_speed=sim.wind_speed,
                 wind_dir=np.asarray(sim.wind_dir, dtype=float), pupil=sim.pupil, pupil_mode=sim.pupil_mode,
                 powerspec_per_layer=sim.powerspec_per_layer, temporal_logamp_powerspec=sim.temporal_logamp_powerspec,
                 pixel_shifts=sim.pixel_shifts, logamp=sim.logamp.copy(), r=res._r, phs_last_chunk=sim.phs.copy(),
                 fx_axis_t=sim.freq.temporal.fx_axis, fy_axis_t=sim.freq.temporal.fy_axis, fabs_t=sim.freq.temporal.fabs)
        save(name, note, True, **d)
    cap_t("temporal_npxls100", base_params(TEMPORAL=True, NITER=16, NCHUNKS=4, DT=0.004, NPXLS=100),
          "TEMPORAL on a 100^2 grid: 16 steps of 4 ms")
    h, cn2, w = turbulence_models.HV57_Bufton_profile(4)
    p = dict(fast.conf.DEFAULTS)
    p.update({"NPXLS": 1000, "DX": 0.01, "NITER": 8, "NCHUNKS": 2, "TEMPORAL": False, "FFTW": True, "SEED": 3, "W0": "opt",
              "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": [0, 90, 180, 270], "ZENITH_ANGLE": 55,
              "DSUBAP": 0.1, "LOGLEVEL": "ERROR", "H_SAT": 36e6, "AO_MODE": "AO", "ALIAS": True})
    capture_sim("big_ao_1000", p, "BASELINE geometry on NPXLS 1000 (AO + ALIAS), NITER 8", full=False, stride=20)


def main():
    os.makedirs(OUT, exist_ok=True)
    print("capturing into", OUT)
    only = [a for a in sys.argv if a.startswith("--only-")]
    if only:
        # add / refresh one family; MANIFEST lines are appended by hand
        if only[0].startswith("--only-e2e="):
            e2e(only[0].split("=", 1)[1].split(","))
        else:
            {"--only-temporal": temporal, "--only-mean-irradiance": mean_irradiance, "--only-stat-ref": stat_ref, "--only-stat-ref-1024": stat_ref_1024, "--only-stat-ref-256b": stat_ref_256b,
             "--only-comms": comms_metrics, "--only-big2048": big2048, "--only-big-modes": big_modes,
             "--only-zenith": zenith, "--only-numpy-branch": numpy_branch, "--only-decimal": decimal, "--only-big-seeds": big_seeds,
             "--only-standin-free": standin_free}[only[0]]()
        for name, size, st, note in MANIFEST:
            print(f"| {name}.npz | {size} | {st} | {note} |")
        return
    kat_fft()
    kat_detector()
    standin_free()
    kat_small()
    e2e()
    temporal()
    mean_irradiance()
    stat_ref()
    stat_ref_256b()
    comms_metrics()
    p = default_cfg()
    numpy_branch()
    if "--no-big" not in sys.argv:
        big(p)
        big2048()            # add --with-4096 for the 4096^2 fixture (70 s of reference init)
        big_modes()
        zenith()
        decimal()
        big_seeds()
        stat_ref_1024()
    with open(os.path.join(OUT, "MANIFEST.md"), "w") as f:
        f.write("# Golden fixtures captured from the reference (tools/capture_golden/capture.py)\n\n")
        f.write(f"numpy {np.__version__}; reference snapshot /root/reference (2025-04-04).\n")
        f.write("'stand-in dependent' = values pass through our aotools stand-ins "
                "(tools/capture_golden/shims/README.md).\n\n")
        f.write("| fixture | bytes | stand-in dependent | content |\n|---|---|---|---|\n")
        for name, size, st, note in MANIFEST:
            f.write(f"| {name}.npz | {size} | {st} | {note} |\n")


if __name__ == "__main__":
    main()
