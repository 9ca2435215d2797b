"""GPU: the multi-GPU paths end to end on a 1-GPU box, without torch.

  * two plain processes (rendezvous through the launcher's environment) sharing device 0 run one sharded
    `Fast.run()`: the in-library RCCL clique cannot form with two ranks on one GPU, so both ranks must fall back to
    the host exchange TOGETHER -- the sharding, seeding, fall-back and reassembly logic is what is under test;
  * one process, two handles on two threads (`GPU_DEVICES: [0, 0]`): bit-identical to the unsharded vector;
  * `bench.py --gpus 2` on a box with one GPU refuses to run instead of measuring one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
import fast_amd
from fast_amd import _lib, multi

pytestmark = pytest.mark.gpu


def _launch(world, port, env_extra):
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", FASTMC_RCCL_TIMEOUT="60")
        env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o[-3000:] + e[-3000:]
    return outs[0][0]


@pytest.mark.parametrize("noseed,coherent,norccl", [("", "", ""), ("1", "", ""), ("", "1", ""), ("", "", "1")])
def test_sharded_run_two_ranks_one_gpu(noseed, coherent, norccl):
    out = _launch(2, 29577, {"NOSEED": noseed, "COHERENT": coherent, "FASTMC_DISABLE_RCCL": norccl})
    assert "GPU DIST OK 2 host" in out


def _params(**over):
    h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
    p = {"NPXLS": 512, "DX": 0.01, "NITER": 600, "NCHUNKS": 4, "SEED": 33, "LOGLEVEL": "ERROR", "D_GROUND": 0.4,
         "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": np.array([0., 90., 180., 270.]), "DSUBAP": 0.1}
    p.update(over)
    return p


@pytest.mark.parametrize("coherent", [False, True])
def test_two_handles_two_threads_equal_one_handle(coherent):
    one = fast_amd.Fast(_params(GPU_DEVICE=0, COHERENT=coherent))
    want = one.run()._r
    sim = fast_amd.Fast(_params(GPU_DEVICES=[0, 0], COHERENT=coherent))
    got = sim.run()._r
    assert sim._group.world == 2 and sim._group.exchange.startswith("host")
    assert np.array_equal(got, want)
    # the device-side reductions see the assembled vector, not one shard
    assert sim.histogram(-60, 10, 70).sum() == 600
    st = sim.result_stats()
    pw = np.abs(want) ** 2 if coherent else want
    assert st["n"] == 600 and st["mean"] == pytest.approx(pw.mean(), rel=1e-12)
    # three shards of unequal size
    sim3 = fast_amd.Fast(_params(GPU_DEVICES=[0, 0, 0], COHERENT=coherent, NITER=500, NCHUNKS=2))
    assert np.array_equal(sim3.run()._r, fast_amd.Fast(_params(GPU_DEVICE=0, COHERENT=coherent, NITER=500, NCHUNKS=2)).run()._r)


def test_device_group_histogram_and_rccl_refusal():
    grp = multi.DeviceGroup(256, 40, "f64", [0, 0])
    assert grp.exchange.startswith("host") and "one device per rank" in grp.exchange
    ps = np.full((256, 256), 1e-3)
    ps[128, 128] = 0.0
    W = np.ones((40, 40))
    grp.set_spectrum(ps, 0.25)
    grp.set_pupil(W, 108, 0.01)
    out = grp.run(9, 0, 300, None, 0.01, False, hist_range=(-40.0, 10.0, 100))
    h = _lib.Handle(256, 40, "f64", 0)
    h.set_spectrum(ps, 0.25)
    h.set_pupil(W, 108, 0.01)
    assert np.array_equal(out, h.run(9, 0, 300, None, 0.01))
    assert np.array_equal(grp.last_hist, h.histogram(-40.0, 10.0, 100)) and grp.last_hist.sum() == 600
    with pytest.raises(fast_amd.FastMCError, match="RCCL exchange requested"):
        multi.DeviceGroup(256, 40, "f64", [0, 0], exchange="rccl")
    # a clique of ONE device initialises (ncclCommInitAll, world 1) and the grouped exchange returns the run itself
    _lib.comm_init_all([h])
    assert h.comm_world() == (1, 0)
    allp, hist = _lib.comm_gather_all([h], 600, (-40.0, 10.0, 100))
    assert np.array_equal(allp, out) and np.array_equal(hist, grp.last_hist)
    h.comm_destroy()
    assert h.comm_world() == (0, -1)


def test_bench_refuses_more_gpus_than_visible():
    n = _lib.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "visible" in (r.stderr + r.stdout)
    assert not any(l.startswith("{") and '"n_gpus"' in l for l in r.stdout.splitlines())


def test_bench_two_threads_on_one_gpu_reports_what_ran():
    """FASTMC_BENCH_DEVICES lets the single-process multi-device bench path run on a 1-GPU box (both workers on device 0):
    the JSON must say 2 workers on 1 distinct device and name the host exchange."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FASTMC_BENCH_DEVICES"] = "0,0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["workers"] == 2 and line["config"]["result_exchange"].startswith("host")
    assert line["config"]["histogram_total"] == 2 * 10000


def test_rccl_transport_code_path_with_a_world_of_one():
    """The one-process-per-GPU RCCL path (unique id -> ncclCommInitRank -> all-gather + all-reduce on the device buffers)
    end to end with a single rank, the only world size RCCL accepts on a 1-GPU box: transport decision, gather,
    gather + histogram, histogram alone, sharded run, teardown."""
    from fast_amd import dist, rendezvous
    h = _lib.Handle(256, 40, "f64", 0)
    if h.comm_world()[0]:
        h.comm_destroy()
    ps = np.full((256, 256), 1e-3)
    ps[128, 128] = 0.0
    h.set_spectrum(ps, 0.25)
    h.set_pupil(np.ones((40, 40)), 108, 0.01)
    rdzv = rendezvous.Rendezvous(0, 1, "unix", "fastmc-test-world1")
    dist._TRANSPORT.pop(h.device, None)
    try:
        tr = dist.make_transport(h, rdzv, rccl_timeout=60)
        assert tr.name == "rccl" and h.comm_world() == (1, 0)
        out = h.run(4, 0, 200, None, 0.01)
        parts = tr.gather(out, h)
        assert len(parts) == 1 and np.array_equal(parts[0], out)
        allp, hist = tr.gather_with_hist(out, h, (-40.0, 10.0, 64))
        assert np.array_equal(allp[0], out) and np.array_equal(hist, h.histogram(-40.0, 10.0, 64)) and hist.sum() == 400
        assert np.array_equal(tr.device_hist(h, (-40.0, 10.0, 64)), hist)
        full = dist.run_sharded(200, lambda r0, n: h.run(4, r0, n, None, 0.01), tr, h)
        assert np.array_equal(full, out)
        coh = dist.run_sharded(200, lambda r0, n: h.run(4, r0, n, None, 0.01, True), tr, h)
        assert np.iscomplexobj(coh) and np.allclose(np.abs(coh) ** 2, out, rtol=1e-12)
    finally:
        dist._TRANSPORT.pop(h.device, None)
        h.comm_destroy()
    assert h.comm_world() == (0, -1)
