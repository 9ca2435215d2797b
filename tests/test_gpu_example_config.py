"""The reference's shipped example configuration (the DATA of its test/test_params.py, rebuilt here key by key) and the one-key
variants its smoke script walks through, run through the drop-in `import fast` alias on the GPU -- with this repo's own checks:
what the golden-fixture parity tests do not already cover.  Every variant: grid and window sizes, result type and length,
strictly positive finite powers, the result's unit conversions, and -- for the Monte-Carlo (non-TEMPORAL) variants -- agreement
of the sample mean with the Fourier model of the same object (`compute_mean_irradiance`, fast/fast.py:736-761, times the
log-amplitude factor exp(2 sigma_chi^2)) within its sampling error."""
import os

import numpy as np
import pytest

import fast   # the alias package of this repo -> fast_amd

pytestmark = pytest.mark.gpu


def shipped_example(**over):
    heights, cn2, wind = fast.turbulence_models.HV57_Bufton_profile(4)
    p = dict(NPXLS="auto", DX=0.01, NITER=100, SUBHARM=False, FFTW=False, FFTW_THREADS=1, NCHUNKS=10, TEMPORAL=True, DT=0.001,
             LOGFILE=None, LOGLEVEL="ERROR", SEED=None, WVL=1550e-9, POWER=1, W0="opt", D_GROUND=0.8, OBSC_GROUND=0, D_SAT=0.1,
             OBSC_SAT=0, AXICON=False, SMF=True, H_SAT=36e6, L_SAT=None, H_TURB=heights, CN2_TURB=cn2, WIND_SPD=wind,
             WIND_DIR=[0, 90, 180, 270], L0=np.inf, l0=1e-6, ZENITH_ANGLE=55, PROP_DIR="up", DTHETA=[4, 0], TRANSMISSION=1,
             AO_MODE="AO", DSUBAP=0.1, TLOOP=0.001, TEXP=0.001, ALIAS=True, NOISE=0, MODAL=False, MODAL_MULT=1, ZMAX=None,
             COHERENT=False, MODULATION=None, EsN0=None, GPU_DEVICE=0)
    p.update(over)
    if os.environ.get("FASTMC_EXAMPLE_TESTS_ON_HOST"):       # dry run of this file on a box without a GPU (numpy host path; device-only
        p.update(GPU_FALLBACK=True, GPU_DEVICE=None)         # checks skip themselves): how the assertions were developed
    return p


# (one-key changes of the reference's smoke script, as data) -> the grid this repo must end up on.  The shipped example auto-sizes
# to 164 (fast/fast.py:167-189) and so does this repo by default (GPU_ROUND_NPXLS False since round 6: a drop-in keeps the reference's
# grid; chirp-z kernels); GPU_ROUND_NPXLS 'auto' / True round a device-generator Monte-Carlo run up to the next fast-kernel size.
VARIANTS = [
    (dict(FFTW=True), 164), (dict(TEMPORAL=False), 164), (dict(TEMPORAL=False, GPU_ROUND_NPXLS="auto"), 192), (dict(SUBHARM=True, TEMPORAL=False), 164), (dict(OBSC_GROUND=0.1), 164),
    (dict(OBSC_SAT=0.05), 164), (dict(W0=0.1, AXICON=True, OBSC_GROUND=0.1), 164), (dict(L0=25), None), (dict(PROP_DIR="down"), 164),
    (dict(AO_MODE="NOAO"), None), (dict(AO_MODE="TT"), 164), (dict(NOISE=1), 164), (dict(MODAL=True), 164),
    (dict(TEMPORAL=False, GPU_PRECISION="f32"), 164), (dict(TEMPORAL=False, GPU_PRECISION="f32", GPU_ROUND_NPXLS=True), 192), (dict(TEMPORAL=False, GPU_RNG="host"), 164), (dict(TEMPORAL=False, NPXLS=256), 256),
]


def check_result(sim, niter, complex_result=False):
    res = sim.result
    raw = np.asarray(res._r)
    assert raw.shape == (niter,) and raw.dtype == (np.complex128 if complex_result else np.float64)
    assert np.isfinite(raw).all()
    if complex_result:          # COHERENT: complex amplitudes relative to the diffraction limit; the unit conversions are for powers
        assert (np.abs(raw) > 0).all()
        return
    assert np.array_equal(np.asarray(sim.I), np.asarray(res.power))      # `I`: the powers, kept for old callers (fast/fast.py:137)
    power = np.asarray(res.power)
    assert power.shape == (niter,) and np.isfinite(power).all() and (power > 0).all()
    # unit conversions of FastResult (fast/fast.py:949-983): dB relative to the diffraction limit, dB of the launched power, dBm
    np.testing.assert_allclose(res.dB_rel, 10 * np.log10(raw), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(10 ** (np.asarray(res.dB_abs) / 10), power, rtol=1e-11)
    np.testing.assert_allclose(10 ** (np.asarray(res.dBm) / 10) * 1e-3, power, rtol=1e-11)
    np.testing.assert_allclose(power, raw * sim.diffraction_limit, rtol=1e-12)


@pytest.mark.parametrize("change, grid", VARIANTS, ids=["-".join(f"{k}={v}" for k, v in c.items()) for c, _ in VARIANTS])
def test_every_variant_of_the_shipped_example(change, grid):
    sim = fast.Fast(shipped_example(**change))
    sim.run()
    assert sim.Npxls_pup == 82                                    # ceil(D_GROUND / DX) + 2 (fast/fast.py:211)
    if grid is not None:
        assert sim.Npxls == grid
    assert sim.Npxls % 2 == 0 and sim.Npxls >= 2 * sim.Npxls_pup
    check_result(sim, 100)
    assert sim.phs.shape[-2:] == (82, 82) and np.isfinite(sim.phs).all()
    assert 0 < sim.r0 < 1 and sim.theta0 > 0 and sim.tau0 > 0


def test_the_shipped_example_itself_and_its_file_form(tmp_path):
    """As shipped (TEMPORAL on, NPXLS auto -> 164), from the dict and from a .py file that defines `p` (fast/conf.py:92-101);
    keys the file leaves out are filled from the defaults."""
    sim = fast.Fast(shipped_example(SEED=12))
    sim.run()
    assert (sim.Npxls, sim.Npxls_pup) == (164, 82)
    check_result(sim, 100)
    cfg = tmp_path / "params.py"
    cfg.write_text("import fast\nh, cn2, w = fast.turbulence_models.HV57_Bufton_profile(4)\n"
                   "p = dict(NPXLS='auto', DX=0.01, NITER=100, NCHUNKS=10, H_TURB=h, CN2_TURB=cn2, WIND_SPD=w, WIND_DIR=[0, 90, 180, 270],\n"
                   "         D_GROUND=0.8, ZENITH_ANGLE=55, DSUBAP=0.1, LOGLEVEL='ERROR', SEED=12, TEMPORAL=True, DT=0.001)\n")
    parsed = fast.conf.ConfigParser(str(cfg)).config
    assert parsed["AO_MODE"] == "AO" and parsed["NITER"] == 100 and parsed["GPU_RNG_PRECISION"] == "auto"
    from_file = fast.Fast(str(cfg))
    from_file.run()
    np.testing.assert_allclose(from_file.result.power, sim.result.power, rtol=1e-12)      # same seed, same series


@pytest.mark.parametrize("change", [dict(), dict(L0=25), dict(AO_MODE="NOAO"), dict(AO_MODE="TT"), dict(PROP_DIR="down")],
                         ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()) or "AO")
def test_sample_mean_agrees_with_the_fourier_model(change):
    """Monte Carlo against the analytic path of the SAME object: E[power] = compute_mean_irradiance() exp(2 sigma_chi^2) (the
    phase enters through the optical transfer function, the log-amplitude as an independent log-normal factor): 4000 iterations,
    five standard errors + 1 % for what the finite window leaves out."""
    sim = fast.Fast(shipped_example(TEMPORAL=False, NITER=4000, NCHUNKS=40, SEED=5, **change))
    sim.run()
    power = np.asarray(sim.result.power)
    model = float(sim.compute_mean_irradiance()) * np.exp(2 * float(sim.logamp_var))
    stderr = power.std() / np.sqrt(power.size)
    assert abs(power.mean() - model) < 5 * stderr + 0.01 * model
    assert abs(sim.result.avg_power_dB_rel - 10 * np.log10(model / sim.diffraction_limit)) < 0.25
    assert sim.result.scintillation_index == pytest.approx(np.var(power / power.mean()), rel=1e-9)


def test_coherent_results_and_the_path_length_key():
    sim = fast.Fast(shipped_example(COHERENT=True, TEMPORAL=False, SEED=2))
    sim.run()
    assert np.asarray(sim.I).dtype == np.complex128
    check_result(sim, 100, complex_result=True)
    incoherent = fast.Fast(shipped_example(TEMPORAL=False, SEED=2))
    incoherent.run()
    np.testing.assert_allclose(np.abs(np.asarray(sim.result._r)) ** 2, incoherent.result._r, rtol=1e-9)      # same draws, |a|^2
    assert fast.Fast(shipped_example(L_SAT=500e3)).L == 500e3                                         # explicit path length wins


def test_mean_irradiance_on_and_off_axis():
    sim = fast.Fast(shipped_example())
    on = sim.compute_mean_irradiance()
    psf = sim.compute_mean_irradiance(onaxis=False)
    assert np.isscalar(on) or np.ndim(on) == 0
    assert psf.shape == (sim.Npxls, sim.Npxls) and np.isfinite(psf).all()
    assert 0 < float(on) <= sim.diffraction_limit * 1.0001
    # the on-axis value is the integral of the transfer function = the focal-plane image at the origin (centre pixel)
    c = sim.Npxls // 2
    assert psf[c, c] == psf.max() and psf[c, c] > 0


def test_save_and_load_round_trip(tmp_path):
    sim = fast.Fast(shipped_example(TEMPORAL=False, SEED=4))
    sim.run()
    f = str(tmp_path / "out.fits")
    sim.save(f)
    back = fast.load(f)
    np.testing.assert_array_equal(back.power, sim.result.power)
    np.testing.assert_allclose(back.dB_rel, sim.result.dB_rel, rtol=1e-12)
    assert back.hdr["NPXLS"] == 164 and back.hdr["AO_MODE"] == "AO" and back.hdr["SEED"] == 4     # the reference's auto-sized grid (GPU_ROUND_NPXLS False)


def test_error_rate_integrals_on_resident_results_equal_the_host_reductions():
    """fast.comms.ber_ook / ber_qam (fast/comms.py:171-262) given the simulation object reduce the results where they are (on the
    device); given the power vector they reduce on the host: same numbers.  Without samples: the textbook AWGN values."""
    from math import erfc, sqrt
    sim = fast.Fast(shipped_example(SEED=9))
    sim.run()
    power = sim.result.power
    for fn, args in ((fast.comms.ber_ook, (10,)), (fast.comms.ber_qam, (4, 10)), (fast.comms.ber_qam, (16, 14))):
        host = fn(*args, samples=power) if fn is fast.comms.ber_qam else fn(*args, power)
        dev = fn(*args, samples=sim) if fn is fast.comms.ber_qam else fn(*args, sim)
        assert np.isfinite(host) and 0 < host < 0.5
        assert dev == pytest.approx(host, rel=1e-11)
    assert fast.comms.ber_ook(10) == pytest.approx(0.5 * erfc(sqrt(10.0) / sqrt(2)), rel=1e-9)          # Q(sqrt(Eb/N0))
    assert 0 < fast.comms.ber_qam(4, 10) < fast.comms.ber_qam(16, 10) < 0.5
