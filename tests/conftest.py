import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def params_from_json(s):
    """Inverse of tools/capture_golden/capture.py:params_to_json."""
    raw = json.loads(str(s))
    out = {}
    for k, v in raw.items():
        if isinstance(v, dict) and "__ndarray__" in v:
            out[k] = np.array(v["__ndarray__"], dtype=float)
        elif isinstance(v, dict) and "__float__" in v:
            out[k] = float(v["__float__"])
        else:
            out[k] = v
    out.setdefault("GPU_ROUND_NPXLS", False)     # captured configs are compared with the reference on ITS grid
    return out


E2E_CASES = ["ao_alias", "noao", "noao_L0", "tt", "noise", "noalias_noise", "modal", "modal_zmax",
             "lgsao", "subharm", "subharm_ao", "coherent", "down", "obsc", "axicon", "w0fixed",
             "lsat_aniso", "oddNp", "autosize", "oddN", "npxls100", "npxls150", "npxls200"]


@pytest.fixture(scope="session")
def golden():
    return load_golden


def run_with_explicit_pupil(name, **extra):
    """A `fast_amd.Fast` built from the fixture's config, then -- as a user of the reference would, by assigning the object's
    attributes -- given the fixture's explicit pupil weights and a unit pupil filter, its spectra recomputed, run.  Returns
    (fixture, sim, result).  Shared by the host-path test (CPU) and the GPU parity test of the stand-in-free fixtures."""
    import numpy as np
    import fast_amd
    g = load_golden(name)
    p = params_from_json(g["params_json"])
    p.update(extra)
    sim = fast_amd.Fast(p)
    sim.pupil = g["W"]                                  # the product pupil x mode is what the detector reads (fast/fast.py:649)
    sim.pupil_mode = np.ones_like(g["W"])
    sim.pupil_filter = 1.0
    sim.compute_powerspec()
    return g, sim, sim.run()


def check_explicit_pupil_run(g, sim, res, rtol=1e-9):
    import numpy as np
    peak = np.abs(g["powerspec"]).max()
    np.testing.assert_allclose(sim.powerspec, g["powerspec"], rtol=1e-10, atol=1e-13 * peak)
    for k in ("logamp_var", "phs_var", "fitting_error", "aniso_servo_error", "alias_error", "noise_error"):
        np.testing.assert_allclose(getattr(sim, k), g[k], rtol=1e-9, atol=1e-300, err_msg=k)
    np.testing.assert_allclose(sim.logamp, g["logamp"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(res._r, g["r"], rtol=rtol)
    assert np.asarray(res._r).dtype == g["r"].dtype
    np.testing.assert_allclose(sim.phs, g["phs_last_chunk"], rtol=0, atol=1e-10 * np.abs(g["phs_last_chunk"]).max())
