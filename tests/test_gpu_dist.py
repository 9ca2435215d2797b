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


def test_sharded_run_two_ranks_with_a_stalled_device_exchange():
    """FASTMC_TEST_STALL_GATHER=1: both ranks believe the RCCL clique is up and fastmc_comm_gather never answers.  The
    deadline passes, the verdict is collective, both ranks abort (fastmc_comm_abort), fetch their own vector (fastmc_wait)
    and finish on the host sockets: same vector as the unsharded run, no hang, exit code 0."""
    out = _launch(2, 29579, {"FASTMC_TEST_STALL_GATHER": "1", "FASTMC_EXCHANGE_TIMEOUT": "2"})
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


def test_eight_workers_on_one_device_equal_one_worker_at_configs1_size(monkeypatch):
    """BASELINE configs[1] geometry (1024^2, Np = 82), eight worker handles on device 0 through the threads path, 10 000
    iterations each: the assembled vector and the histogram are bit-identical to ONE handle computing all 80 000.  Then the
    same group with a device exchange that never answers (FASTMC_TEST_STALL_GATHER=1): the deadline passes, the clique is
    aborted and the step still returns the identical vector, from the host path, which the group keeps from then on."""
    g = __import__("conftest").load_golden("big_noao_L0_1024")
    p = __import__("conftest").params_from_json(g["params_json"])
    p.update({"NITER": 80000, "NCHUNKS": 8, "SEED": 3, "GPU_RNG": "device", "LOGLEVEL": "ERROR"})
    one = fast_amd.Fast(dict(p, GPU_DEVICE=0))
    want = one.run()._r
    hist1 = one.histogram(-60.0, 10.0, 4096)
    assert one.Npxls == 1024 and one.Npxls_pup == 82 and want.shape == (80000,) and (want > 0).all()
    eight = fast_amd.Fast(dict(p, GPU_DEVICES=[0] * 8))
    got = eight.run()._r
    assert eight._group.world == 8 and eight._group.exchange.startswith("host") and np.array_equal(got, want)
    assert np.array_equal(eight.histogram(-60.0, 10.0, 4096), hist1)
    busy = [t["rows_ms"] + t["cols_ms"] for t in eight._group.last_timing()]
    assert len(busy) == 8 and min(busy) > 0
    monkeypatch.setenv("FASTMC_TEST_STALL_GATHER", "1")
    monkeypatch.setenv("FASTMC_EXCHANGE_TIMEOUT", "3")
    stalled = fast_amd.Fast(dict(p, GPU_DEVICES=[0] * 8))
    assert stalled._group.exchange == "rccl" and stalled._group.rccl_ranks == 8        # what the fault injection pretends
    import time
    t0 = time.perf_counter()
    got2 = stalled.run()._r
    assert 2.5 < time.perf_counter() - t0 < 60
    assert np.array_equal(got2, want)
    assert stalled._group.exchange.startswith("host (RCCL exchange given up: no answer within 3 s") and stalled._group.last_exchange == "host"
    monkeypatch.delenv("FASTMC_TEST_STALL_GATHER")
    t0 = time.perf_counter()
    assert np.array_equal(stalled.run()._r, want) and time.perf_counter() - t0 < 2.5      # no second deadline: host path kept
    from fast_amd import dist
    time.sleep(0.2)
    assert not dist.stuck_threads()


def test_async_run_and_wait_equal_the_blocking_run():
    """fastmc_run_async + fastmc_wait: same vector as fastmc_run, timing read after the wait, statistics / histogram ordered
    behind the kernels on the stream; a second wait fetches the same results again."""
    h = _lib.Handle(256, 40, "f64", 0)
    ps = np.full((256, 256), 1e-3)
    ps[128, 128] = 0.0
    h.set_spectrum(ps, 0.25)
    h.set_pupil(np.ones((40, 40)), 108, 0.01)
    want = h.run(9, 5, 300, None, 0.01)
    h.run_async(9, 5, 300, 0.01)
    hist = h.histogram(-40.0, 10.0, 50)                 # stream-ordered behind the kernels
    got = h.wait()
    assert np.array_equal(got, want) and hist.sum() == 600 and np.array_equal(h.wait(), want)
    t = h.last_timing()
    assert t["rows_launches"] >= 1 and t["rows_ms"] > 0 and t["cols_ms"] > 0
    h.run_async(9, 5, 300, 0.01, coherent=True)
    coh = h.wait()
    assert np.iscomplexobj(coh) and np.allclose(np.abs(coh) ** 2, want, rtol=1e-12)
    h.run_async(9, 0, 10, 0.01)                         # never waited for: the next call tidies up
    assert np.array_equal(h.run(9, 5, 300, None, 0.01), want)
    # abort without a communicator is a no-op; the exchange time of a handle that never exchanged is 0
    h.comm_abort()
    assert h.comm_world() == (0, -1) and h.last_exchange_ms() == 0.0


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
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["workers"] == 2 and line["config"]["result_exchange"].startswith("host")
    assert line["config"]["histogram_total"] == 2 * 10000 and line["config"]["rccl_ranks"] == 0
    assert line["exchange"]["steps_host"] == 2 and line["exchange"]["steps_rccl"] == 0 and line["exchange"]["host_ms_per_step"] >= 0
    busy = line["pipeline"]["gpu_busy_ms_per_step"]
    assert len(busy["per_worker"]) == 2 and 0 < busy["min_worker"] <= busy["max_worker"]


def test_bench_eight_workers_with_a_stalled_exchange_prints_its_line_and_says_what_happened():
    """bench.py --gpus 8 (eight worker threads on device 0) with FASTMC_TEST_STALL_GATHER=1: the first step's exchange never
    answers; bench.py must not hang -- it aborts, goes on with the host exchange and the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"FASTMC_BENCH_DEVICES": ",".join(["0"] * 8), "FASTMC_TEST_STALL_GATHER": "1", "FASTMC_EXCHANGE_TIMEOUT": "3"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras", "--no-sustained"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["workers"] == 8 and line["config"]["rccl_ranks"] == 0
    assert line["config"]["result_exchange"].startswith("host (RCCL exchange given up: no answer within 3 s")
    assert line["exchange"]["steps_host"] == 2 and line["config"]["histogram_total"] == 8 * 10000


def test_bench_multi_gpu_line_carries_config3_and_the_config5_sweep_over_all_workers():
    """one `bench.py --gpus N` line (one process, N worker threads -- here four on device 0) carries the weak-scaling
    headline AND configs[3] over all N workers (strong scaling, gather + histogram) AND the 32-angle configs[4] sweep dealt 32 / N per
    worker, each with its exchange named."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FASTMC_BENCH_DEVICES"] = "0,0,0,0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-sustained", "--no-f32-draw-pass", "--no-host-cost-pass"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "weak" and line["config"]["workers"] == 4
    x = line["extras_multi_gpu"]
    assert "error" not in x, x
    c3, c5 = x["config3_2048_100k_all_gpus"], x["config5_zenith_scan_32x4096_all_gpus"]
    assert c3["workers"] == 4 and c3["histogram_total"] == 100000 and c3["scaling"] == "strong" and c3["iterations_per_s"] > 1e4
    assert c3["result_exchange"].startswith("host") and c3["rccl_ranks"] == 0          # four workers share one device here
    assert c5["samples"] == 32 and c5["workers"] == 4 and c5["samples_per_worker"] == 8 and c5["iterations_per_s_end_to_end"] > 1e4
    assert c5["samples_by_device_index"] == {"0": 32}
    assert c5["mean_dB_rel_first_last"][0] > c5["mean_dB_rel_first_last"][1]


def test_bench_two_ranks_line_carries_the_multi_gpu_extras():
    """The launcher form (one process per GPU: how the driver starts bench.py for N > 1), two ranks on the one GPU of a test box:
    configs[3] sharded over the ranks (GPU_SHARD auto) and the configs[4] sweep dealt angles[rank::world], records gathered."""
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29591", FASTMC_BENCH_DEVICE="0", FASTMC_DISABLE_RCCL="1", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                                       "--no-sustained", "--no-f32-draw-pass", "--no-host-cost-pass"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o[-3000:] + e[-3000:]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert line["config"]["workers"] == 2 and "processes" in line["config"]["launch"]
    x = line["extras_multi_gpu"]
    assert "error" not in x, x
    assert x["config3_2048_100k_all_gpus"]["histogram_total"] == 100000 and x["config3_2048_100k_all_gpus"]["result_exchange"] == "host"
    assert x["config5_zenith_scan_32x4096_all_gpus"]["samples"] == 32 and x["config5_zenith_scan_32x4096_all_gpus"]["samples_per_worker"] == 16
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]            # rank 0 alone prints the line


def test_bench_config3_strong_scaling_workload():
    """--workload config3: BASELINE configs[3] (2048^2, 100 000 iterations per step in total) split over the workers."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FASTMC_BENCH_DEVICES"] = "0,0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "config3",
                        "--no-cpu-baseline", "--no-extras", "--no-sustained"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and "2048^2" in line["metric"] and line["config"]["iters_per_step_per_gpu"] == 50000
    assert line["config"]["histogram_total"] == 100000 and abs(line["value"] - 100000 / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]


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


def test_abort_while_a_real_exchange_holds_the_handle():
    """the stall injection used to return before touching RCCL.  FASTMC_TEST_STALL_GATHER=2 blocks AFTER the
    collectives of a real communicator (world of one: all a 1-GPU box allows) are on the stream, holding the handle's lock:
    the deadline passes, the caller aborts the communicator from another thread, and only then touches the handle -- the
    library's per-handle lock and the bounded join keep the two apart; the device's own vector is intact."""
    code = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
os.environ["FASTMC_EXCHANGE_TIMEOUT"] = "1.0"
from fast_amd import _lib, dist, rendezvous
h = _lib.Handle(256, 40, "f64", 0)
ps = np.full((256, 256), 1e-3); ps[128, 128] = 0.0
h.set_spectrum(ps, 0.25); h.set_pupil(np.ones((40, 40)), 108, 0.01)
rdzv = rendezvous.Rendezvous(0, 1, "unix", "fastmc-test-stall2")
tr = dist.make_transport(h, rdzv, rccl_timeout=60)
assert tr.name == "rccl" and h.comm_world() == (1, 0)
want = h.run(4, 0, 100, None, 0.01)
os.environ["FASTMC_TEST_STALL_GATHER"] = "2"
t0 = time.perf_counter()
out, hist, info = dist.step_sharded(h, tr, 4, 0, 100, 0.01, False, (-40.0, 10.0, 64))
dt = time.perf_counter() - t0
assert 0.9 < dt < 20, dt
assert info["exchange"] == "host" and tr.name == "host" and "given up" in tr.why
assert np.array_equal(out, want) and hist.sum() == 200
assert h.comm_world() == (0, -1) and not dist.stuck_threads()
# the handle works on: a blocking run, and a second sharded step on the host path
os.environ["FASTMC_TEST_STALL_GATHER"] = "0"
assert np.array_equal(h.run(4, 0, 100, None, 0.01), want)
out2, _, info2 = dist.step_sharded(h, tr, 4, 0, 100, 0.01, False, None)
assert np.array_equal(out2, want) and info2["exchange"] == "host"
# a second caller on a busy handle is refused with a message, not left hanging
import threading
os.environ["FASTMC_HANDLE_BUSY_TIMEOUT"] = "0.5"
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout.splitlines(), r.stdout[-2000:] + r.stderr[-3000:]      # (RCCL prints its banner after it)


def test_two_steps_in_flight_equal_one_step_at_a_time():
    """fastmc_run_queued / fastmc_histogram_queued / fastmc_queue_wait: step i + 1 enqueued before step i is collected, on one
    handle and on a group of three handles (host exchange on a 1-GPU box): vectors and histograms identical to the blocking
    calls, timing of each step readable after its wait, slot misuse refused."""
    p = _params(GPU_DEVICE=0, NITER=64, NCHUNKS=1)
    sim = fast_amd.Fast(dict(p))
    h = sim._handle
    lv = float(sim.logamp_var)
    steps = [(0, 40), (40, 40), (80, 24), (104, 40), (144, 8)]
    hr = (-40.0, 10.0, 32)
    want = [(h.run(3, r0, n, None, lv), h.histogram(*hr)) for r0, n in steps]
    got = list(sim._group.run_pipelined(3, steps, lv, False, hr))
    for (v, hh), (wv, wh) in zip(got, want):
        assert np.array_equal(v, wv) and np.array_equal(hh, wh)
    t = h.last_timing()
    assert t["rows_ms"] > 0 and t["rows_launches"] >= 1
    # coherent amplitudes, no histogram
    wantc = [h.run(3, r0, n, None, lv, True) for r0, n in steps[:3]]
    gotc = [v for v, _ in sim._group.run_pipelined(3, steps[:3], lv, True, None)]
    assert all(np.array_equal(a, b) for a, b in zip(gotc, wantc))
    # slots: a busy slot refuses a second step; an idle one has nothing to wait for
    h.run_queued(3, 0, 8, lv, False, slot=0)
    with pytest.raises(_lib.FastMCError, match="still in flight"):
        h.run_queued(3, 8, 8, lv, False, slot=0)
    with pytest.raises(_lib.FastMCError, match="nothing is queued"):
        h.queue_wait(1)
    v, _ = h.queue_wait(0, 16)
    assert np.array_equal(v, h.run(3, 0, 8, None, lv))
    # a blocking run between queued steps is allowed (stream order): results of both intact
    h.run_queued(3, 0, 8, lv, False, slot=1)
    mid = h.run(3, 100, 8, None, lv)
    v, _ = h.queue_wait(1, 16)
    assert np.array_equal(v, h.run(3, 0, 8, None, lv)) and np.array_equal(mid, h.run(3, 100, 8, None, lv))
    # three handles on the one device
    p3 = dict(p, GPU_DEVICES=[0, 0, 0])
    p3.pop("GPU_DEVICE")
    grp = fast_amd.Fast(p3)._group
    steps3 = [(0, 30), (30, 30), (60, 31), (91, 30)]          # 31: unequal shards
    want3 = [(h.run(3, r0, n, None, lv), h.histogram(*hr)) for r0, n in steps3]
    got3 = list(grp.run_pipelined(3, steps3, lv, False, hr))
    for (v, hh), (wv, wh) in zip(got3, want3):
        assert np.array_equal(v, wv) and np.array_equal(hh, wh)


def test_queued_rccl_exchange_with_a_world_of_one():
    """The queued exchange (fastmc_comm_gather_queued) on a real communicator, world of one, through dist.steps_pipelined:
    identical to step_sharded step by step (the fall-back of a failed queued exchange is exercised on the CPU with stand-in
    handles: tests/test_multi_deadline.py)."""
    code = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
os.environ["FASTMC_EXCHANGE_TIMEOUT"] = "1.5"
from fast_amd import _lib, dist, rendezvous
h = _lib.Handle(256, 40, "f64", 0)
ps = np.full((256, 256), 1e-3); ps[128, 128] = 0.0
h.set_spectrum(ps, 0.25); h.set_pupil(np.ones((40, 40)), 108, 0.01)
rdzv = rendezvous.Rendezvous(0, 1, "unix", "fastmc-test-queued")
tr = dist.make_transport(h, rdzv, rccl_timeout=60)
assert tr.name == "rccl"
steps = [(0, 50), (50, 50), (100, 50), (150, 50)]
hr = (-40.0, 10.0, 16)
want = [dist.step_sharded(h, tr, 4, b, n, 0.01, False, hr) for b, n in steps]
got = list(dist.steps_pipelined(h, tr, 4, steps, 0.01, False, hr))
for (v, hh, info), (wv, wh, _) in zip(got, want):
    assert np.array_equal(v, wv) and np.array_equal(hh, wh) and info["exchange"] == "rccl" and info["exchange_device_ms"] >= 0
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout.splitlines(), r.stdout[-2000:] + r.stderr[-3000:]
