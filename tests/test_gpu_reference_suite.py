"""The reference's own smoke suite (test/tests_pytest.py:30-127 of ojdf/fast: build a Fast from the
example configuration, mutate one key, run, assert finiteness) against the GPU implementation,
through the drop-in `import fast` alias.  The example configuration is rebuilt here key by key
(it is data: test/test_params.py of the reference)."""
import numpy
import pytest

import fast   # the alias package of this repo -> fast_amd

pytestmark = pytest.mark.gpu


def example_params():
    h, cn2, w = fast.turbulence_models.HV57_Bufton_profile(4)
    return {
        'NPXLS': "auto", 'DX': 0.01, 'NITER': 100, 'SUBHARM': False, 'FFTW': False, 'FFTW_THREADS': 1,
        'NCHUNKS': 10, 'TEMPORAL': True, 'DT': 0.001, 'LOGFILE': None, 'LOGLEVEL': "ERROR", 'SEED': None,
        'WVL': 1550e-9, 'POWER': 1, 'W0': "opt", 'D_GROUND': 0.8, 'OBSC_GROUND': 0, 'D_SAT': 0.1, 'OBSC_SAT': 0,
        'AXICON': False, 'SMF': True, 'H_SAT': 36e6, 'L_SAT': None, 'H_TURB': h, 'CN2_TURB': cn2, 'WIND_SPD': w,
        'WIND_DIR': [0, 90, 180, 270], 'L0': numpy.inf, 'l0': 1e-6, 'ZENITH_ANGLE': 55, 'PROP_DIR': 'up',
        'DTHETA': [4, 0], 'TRANSMISSION': 1, 'AO_MODE': 'AO', 'DSUBAP': 0.1, 'TLOOP': 0.001, 'TEXP': 0.001,
        'ALIAS': True, 'NOISE': 0, 'MODAL': False, 'MODAL_MULT': 1, 'ZMAX': None, 'COHERENT': False,
        'MODULATION': None, 'EsN0': None, 'GPU_DEVICE': 0,
    }


def run_sim(p):
    sim = fast.Fast(p)
    sim.run()
    assert numpy.isfinite(sim.I).all()
    return sim


# --- turbulence models (tests_pytest.py:12-27)
def test_HV57():
    h = numpy.linspace(0, 20000, 10)
    cn2 = fast.turbulence_models.HV57(h)
    assert len(cn2) == len(h)
    assert cn2.dtype == float


def test_Bufton():
    h = numpy.linspace(0, 20000, 10)
    w = fast.turbulence_models.Bufton_wind(h)
    assert len(w) == len(h)
    assert w.dtype == float


def test_HV57_Bufton():
    h, cn2, w = fast.turbulence_models.HV57_Bufton_profile(10)
    assert len(h) == len(cn2) == len(w) == 10


# --- config parsing (tests_pytest.py:30-32): a .py file that defines `p`
def test_config_default(tmp_path):
    cfg = tmp_path / "test_params.py"
    cfg.write_text("import numpy\nimport fast\nh, cn2, w = fast.turbulence_models.HV57_Bufton_profile(4)\n"
                   "p = {'NPXLS': 'auto', 'DX': 0.01, 'NITER': 100, 'NCHUNKS': 10, 'H_TURB': h, 'CN2_TURB': cn2, 'WIND_SPD': w,\n"
                   "     'WIND_DIR': [0, 90, 180, 270], 'D_GROUND': 0.8, 'ZENITH_ANGLE': 55, 'DSUBAP': 0.1, 'LOGLEVEL': 'ERROR'}\n")
    c = fast.conf.ConfigParser(str(cfg))
    assert c.config['NITER'] == 100 and c.config['AO_MODE'] == 'AO'        # missing keys filled from the defaults
    run_sim(str(cfg))                                                        # Fast accepts the file name, like the reference


def test_sim_default():
    sim = fast.Fast(example_params())
    sim.run()
    assert sim.Npxls == 164 and sim.Npxls_pup == 82
    assert numpy.isfinite(sim.result.power).all()
    assert numpy.isfinite(sim.result.dB_rel).all()
    assert numpy.isfinite(sim.result.dB_abs).all()


@pytest.mark.parametrize("change", [
    {'FFTW': True}, {'TEMPORAL': False}, {'SUBHARM': True, 'TEMPORAL': False}, {'OBSC_GROUND': 0.1}, {'OBSC_SAT': 0.05},
    {'W0': 0.1, 'AXICON': True, 'OBSC_GROUND': 0.1}, {'L0': 25}, {'PROP_DIR': 'down'}, {'AO_MODE': 'NOAO'},
    {'AO_MODE': 'TT'}, {'NOISE': 1}, {'MODAL': True}, {'TEMPORAL': False, 'GPU_PRECISION': 'f32'},
    {'TEMPORAL': False, 'GPU_RNG': 'host'}, {'TEMPORAL': False, 'NPXLS': 256},
], ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()))
def test_sim_variants(change):
    p = example_params()
    p.update(change)
    run_sim(p)


def test_sim_L_SAT():
    p = example_params()
    p['L_SAT'] = 500e3
    assert fast.Fast(p).L == p['L_SAT']


def test_sim_coherent():
    p = example_params()
    p['COHERENT'] = True
    sim = fast.Fast(p)
    sim.run()
    assert sim.I.dtype == complex


def test_sim_mean_irradiance():
    sim = fast.Fast(example_params())
    psf = sim.compute_mean_irradiance()
    assert numpy.isfinite(psf.all())


def test_save_and_load(tmp_path):
    p = example_params()
    p.update({'TEMPORAL': False, 'SEED': 4})
    sim = fast.Fast(p)
    sim.run()
    f = str(tmp_path / "out.fits")
    sim.save(f)
    res = fast.load(f)
    numpy.testing.assert_allclose(res.power, sim.result.power, rtol=1e-15)
    numpy.testing.assert_allclose(res.dB_rel, sim.result.dB_rel, rtol=1e-12)
    assert res.hdr['NPXLS'] == 256 and res.hdr['AO_MODE'] == 'AO' and res.hdr['SEED'] == 4   # auto 164, rounded up (GPU_ROUND_NPXLS 'auto')


# --- error-rate integrals over the result (tests_pytest.py:168-187)
def test_ber_ook():
    p = example_params()
    sim = fast.Fast(p)
    sim.run()
    ber = fast.comms.ber_ook(10, sim.result.power)
    assert numpy.isfinite(ber) and 0 < ber < 0.5
    assert abs(fast.comms.ber_ook(10, sim) - ber) < 1e-12 * ber             # same reduction on the resident results


def test_ber_ook_nosamples():
    ber = fast.comms.ber_ook(10)
    assert numpy.isfinite(ber)


def test_sep_qam():
    p = example_params()
    sim = fast.Fast(p)
    sim.run()
    ber = fast.comms.ber_qam(4, 10, samples=sim.result.power)
    assert numpy.isfinite(ber)


def test_ber_qam_nosamples():
    ber = fast.comms.ber_qam(4, 10)
    assert numpy.isfinite(ber)
