"""CPU: the opt-in host path (`GPU_FALLBACK: True`; fast_amd/hostpath.py) against the reference's own fixtures -- this container
has no GPU, so `fast_amd.Fast` takes it here exactly as a user on a login node would: a warning in the reference's style
(fast/fast.py:107-110), then numpy.  BASELINE configs[0] ("CPU numpy FFT path, plumbing, no GPU") runs as written."""
import logging

import numpy as np
import pytest

from conftest import E2E_CASES, load_golden, params_from_json
import fast_amd
from fast_amd import _lib


def _no_gpu():
    try:
        return _lib.device_count() == 0
    except Exception:
        return True


needs_no_gpu = pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible: GPU_FALLBACK does not fall back")


def test_default_is_to_raise_without_a_gpu():
    if not _no_gpu():
        pytest.skip("a GPU is visible")
    g = load_golden("e2e_ao_alias")
    p = params_from_json(g["params_json"])
    with pytest.raises(Exception):
        fast_amd.Fast(p)


@needs_no_gpu
@pytest.mark.parametrize("case", E2E_CASES + ["default164"])
def test_host_path_reproduces_the_reference(case, caplog):
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    p["GPU_FALLBACK"] = True
    with caplog.at_level(logging.WARNING):
        sim = fast_amd.Fast(p)
    assert sim.backend == "host" and any("falling back to numpy" in m for m in caplog.messages)
    res = sim.run()
    assert res._r.dtype == g["r"].dtype
    np.testing.assert_allclose(res._r, g["r"], rtol=1e-9)
    np.testing.assert_allclose(sim.logamp, g["logamp"], rtol=1e-9, atol=1e-300)
    for k in ("phs_var", "logamp_var", "fitting_error", "aniso_servo_error", "alias_error", "noise_error"):
        if k in g.files:
            np.testing.assert_allclose(getattr(sim, k), g[k], rtol=1e-9, atol=1e-300, err_msg=k)
    if "powerspec" in g.files:
        np.testing.assert_allclose(sim.powerspec, g["powerspec"], rtol=1e-10, atol=1e-13 * np.abs(g["powerspec"]).max())
    if "phs_last_chunk" in g.files and case != "numpy_branch":
        np.testing.assert_allclose(sim.phs, g["phs_last_chunk"], rtol=1e-9, atol=1e-11 * np.abs(g["phs_last_chunk"]).max())
    assert np.isfinite(res.power).all() and np.isfinite(res.dB_rel).all()
    h = sim.histogram(-40.0, 10.0, 16)
    assert h.sum() == res._r.size and sim.result_stats([-3.0])["n"] == res._r.size


@needs_no_gpu
@pytest.mark.parametrize("name", ["temporal_default", "temporal_noao", "temporal_npxls100", "temporal_small"])
def test_host_path_temporal_series(name):
    """BASELINE configs[0]: the reference's shipped test/test_params.py (TEMPORAL, 100 iterations) as written, no GPU."""
    g = load_golden(name)
    p = params_from_json(g["params_json"])
    p["GPU_FALLBACK"] = True
    sim = fast_amd.Fast(p)
    res = sim.run()
    np.testing.assert_allclose(res._r, g["r"], rtol=1e-9)


def test_the_numpy_path_has_no_grid_size_limit():
    """The limit of 8192 (windows of up to 256 pixels beyond 4096) belongs to the GPU kernels; a run that opted into the numpy path
    (GPU_FALLBACK) and gets it has none, like the reference (fast.py:176-211).  Only the set-up is exercised: the O(N^2) power
    spectrum of a 8200^2 grid is not a CPU-suite test."""
    from fast_amd import host
    from conftest import load_golden, params_from_json
    p = params_from_json(load_golden("e2e_ao_alias")["params_json"])
    import fast_amd.conf as conf
    cfg = conf.ConfigParser(dict(p, NPXLS=8200)).config
    with pytest.raises(Exception, match="exceeds the GPU kernels' limit"):
        host.grid_size(cfg, host.atmosphere(cfg))
    dx, N, Np = host.grid_size(cfg, host.atmosphere(cfg), size_limit=False)
    assert N == 8200


@needs_no_gpu
@pytest.mark.parametrize("name", ["e2e_explicit_pupil", "e2e_explicit_pupil_noao"])
def test_assigned_pupil_weights_reproduce_the_stand_in_free_fixtures(name):
    """`sim.pupil`, `sim.pupil_mode`, `sim.pupil_filter` are assignable as on the reference object; with the fixture's explicit
    weights the run reproduces a reference run that had no aotools stand-in between its config and `result._r`."""
    from conftest import run_with_explicit_pupil, check_explicit_pupil_run
    g, sim, res = run_with_explicit_pupil(name, GPU_FALLBACK=True)
    check_explicit_pupil_run(g, sim, res)
    # the shared description of the geometry was not touched: a second object starts from the computed pupil again
    p = params_from_json(g["params_json"])
    p["GPU_FALLBACK"] = True
    other = fast_amd.Fast(p)
    assert other.pupil.shape == g["W"].shape and not np.array_equal(other.pupil * other.pupil_mode, g["W"])
    with pytest.raises(ValueError):
        other.pupil = np.ones((3, 3))
