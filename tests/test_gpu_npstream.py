"""GPU: numpy's Generator.normal stream drawn on the device (fast_amd/csrc/fmc_npstream.h) against numpy itself -- the oracle
here is the library the reference calls (fast/funcs.py:21, 352-365)."""
import numpy as np
import pytest

import fast_amd
from fast_amd import _lib, npnormal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n", [(1, 1), (1, 7), (2, 5000), (3, 16384), (4, 16385), (5, 100_000), (6, 1_048_576), (2 ** 40 + 7, 3_000_001)])
def test_device_array_is_numpys(seed, n):
    h = _lib.Handle(256, 40, "f64", 0)
    rng = np.random.default_rng(seed)
    sw = npnormal.state_words(rng.bit_generator)
    want = rng.normal(0, 1, n)
    got, after, consumed, ovf = h.npstream_normals(sw, n)
    assert ovf == 0
    d = np.flatnonzero(got != want)
    # ocml's log1p / exp against glibc's: a tail or wedge VALUE is the same number (the fast path and the wedge return rabs * wi),
    # only the tail's R + xx can differ in the last bit
    assert d.size <= max(2, n // 200_000) and np.abs(got - want).max() < 1e-15 * 8
    end = rng.bit_generator.state["state"]["state"]
    assert (int(after[1]) << 64) | int(after[0]) == end                  # the stream position is numpy's, word for word
    st0 = {"lo": int(sw[0]), "hi": int(sw[1])}
    assert npnormal.pcg64_advance((st0["hi"] << 64) | st0["lo"], (int(sw[3]) << 64) | int(sw[2]), consumed) == end
    assert n <= consumed < 1.03 * n + 40


def test_consecutive_arrays_continue_the_stream():
    h = _lib.Handle(256, 40, "f64", 0)
    rng = np.random.default_rng(77)
    sw = npnormal.state_words(rng.bit_generator)
    for n in (1000, 20000, 333):
        want = rng.normal(0, 1, n)
        got, after, _, ovf = h.npstream_normals(sw, n)
        assert ovf == 0 and np.abs(got - want).max() < 1e-14
        sw = np.array([after[0], after[1], sw[2], sw[3]], dtype=np.uint64)


# ------------------------------------------------------------------ Fast(config).run() with GPU_RNG 'numpy'
from conftest import E2E_CASES, load_golden, params_from_json


@pytest.mark.parametrize("case", E2E_CASES + ["default164"])
def test_fast_run_reproduces_the_reference_for_its_seed_without_a_host_draw(case):
    """Every end-to-end fixture of the reference (all AO modes, alias, noise, modal, sub-harmonics, coherent, downlink, odd
    grids ...): `result._r`, the log-amplitudes, `phs` of the last chunk, AND the state the module generator is left in -- with
    the draws made on the device."""
    g = load_golden("e2e_" + case)
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "numpy", "GPU_DEVICE": 0})
    sim = fast_amd.Fast(p)
    res = sim.run()
    assert res._r.dtype == g["r"].dtype
    np.testing.assert_allclose(res._r, g["r"], rtol=1e-9)
    np.testing.assert_allclose(sim.logamp, g["logamp"], rtol=1e-9, atol=1e-300)
    if "phs_last_chunk" in g.files and case != "numpy_branch":
        np.testing.assert_allclose(sim.phs, g["phs_last_chunk"], rtol=1e-9, atol=1e-11 * np.abs(g["phs_last_chunk"]).max())
    # the module generator is where the reference's would be: a host-mode run of the same config leaves it in the same state
    after_dev = fast_amd.fast._R.bit_generator.state
    p["GPU_RNG"] = "host"
    sim2 = fast_amd.Fast(p)
    res2 = sim2.run()
    assert fast_amd.fast._R.bit_generator.state["state"] == after_dev["state"]
    np.testing.assert_allclose(res._r, res2._r, rtol=1e-12)


@pytest.mark.parametrize("name", ["cfg1_256", "big_noao_1024", "big_noao_L0_1024", "big_ao_1024", "big_noao_L0_2048", "big_subharm_coherent_down_1024",
                                  "big_ao_1000", "big_noao_1024_s2"])
def test_fast_run_full_size_same_seed_on_the_device(name):
    g = load_golden(name)
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "numpy", "GPU_DEVICE": 0})
    sim = fast_amd.Fast(p)
    np.testing.assert_allclose(sim.run()._r, g["r"], rtol=1e-8)


def test_numpy_mode_falls_back_to_host_draws_when_the_generator_is_not_pcg64(caplog):
    g = load_golden("e2e_ao_alias")
    p = params_from_json(g["params_json"])
    p.update({"GPU_RNG": "numpy", "GPU_DEVICE": 0})
    sim = fast_amd.Fast(p)
    fast_amd.fast._R = np.random.Generator(np.random.Philox(5))
    import logging
    with caplog.at_level(logging.WARNING):
        r = sim.run()._r
    assert any("drawing on the host" in m for m in caplog.messages)
    want_rng = np.random.Generator(np.random.Philox(5))
    assert np.isfinite(r).all() and r.size == g["r"].size
    sim.set_seed(1)


def test_three_pass_form_gives_the_same_stream():
    """Segments too long for a buffer of their own take the three-pass form (classify, scan, emit): force it and compare again --
    also with the scan's in-order composition of the tile maps (taken when a tile's exit offset depends on its entry, which real
    streams never reach)."""
    import os, subprocess, sys
    code = r'''
import numpy as np
from fast_amd import _lib, npnormal
h = _lib.Handle(256, 40, "f64", 0)
for seed, n in ((1, 1), (2, 5000), (4, 16385), (6, 1_048_576), (9, 3_000_001)):
    rng = np.random.default_rng(seed)
    sw = npnormal.state_words(rng.bit_generator)
    want = rng.normal(0, 1, n)
    got, after, consumed, ovf = h.npstream_normals(sw, n)
    assert ovf == 0 and np.abs(got - want).max() < 8e-15, (seed, n, ovf)
    assert (int(after[1]) << 64) | int(after[0]) == rng.bit_generator.state["state"]["state"]
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"FASTMC_NPS_THREEPASS": "1"}, {"FASTMC_NPS_THREEPASS": "1", "FASTMC_NPS_GENERAL_SCAN": "1"}):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout.splitlines(), r.stdout + r.stderr


def test_a_chunk_the_device_gives_up_on_is_redrawn_by_numpy_and_the_run_goes_on():
    """An overflow flag (a normal spanning more than 16 words across a tile edge, ...) has never been raised by a real stream; the
    library can be told to report one (FASTMC_NPS_TEST_OVERFLOW): the run must draw that chunk with numpy from the state at its
    start, continue on the device, and end with the reference's numbers and generator state."""
    import os, subprocess, sys
    code = r'''
import logging, sys
import numpy as np
sys.path.insert(0, "tests")
import fast_amd
from conftest import load_golden, params_from_json
g = load_golden("e2e_ao_alias")
p = params_from_json(g["params_json"])
p.update({"GPU_RNG": "numpy", "GPU_DEVICE": 0, "LOGLEVEL": "WARNING"})
msgs = []
class H(logging.Handler):
    def emit(self, rec): msgs.append(rec.getMessage())
logging.getLogger("fast_amd").addHandler(H())
sim = fast_amd.Fast(p)
r = sim.run()._r
assert any("gave up" in m for m in msgs), msgs
assert np.allclose(r, g["r"], rtol=1e-9, atol=0), np.abs(r / g["r"] - 1).max()
# ... and the module generator stands where a run without the incident leaves it
end = fast_amd.fast._R.bit_generator.state["state"]["state"]
p2 = dict(p, GPU_RNG="host")
fast_amd.Fast(p2).run()
assert fast_amd.fast._R.bit_generator.state["state"]["state"] == end
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, FASTMC_NPS_TEST_OVERFLOW="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout.splitlines(), r.stdout[-2000:] + r.stderr[-3000:]
