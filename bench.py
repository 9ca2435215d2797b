#!/usr/bin/env python3
"""bench.py -- Monte-Carlo iterations/s of the FAST hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        one process drives N GPUs (N handles on N threads)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W       one process per GPU (any launcher that sets
                                                                     RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)

No torch in either form: under a launcher the ranks meet through fast_amd/rendezvous.py (plain sockets) and the
results are exchanged by RCCL inside libfastmc.so (host sockets if RCCL cannot initialise on EVERY rank).  `--gpus N`
with more GPUs than are visible, or a launcher world that differs from N, exits non-zero: the line never reports
devices that did not run.

Workload (BASELINE.json configs[1]): 1024 x 1024 grid, Np = 82 pupil window (D = 0.8 m, DX = 0.01 m), pure von Karman
spectrum (AO_MODE 'NOAO', HV5/7 + Bufton 4-layer profile at 55 deg zenith), 10 000 Monte-Carlo iterations, on-device
generator, float64 transform.  One "step" = that whole 10 000-iteration job on each GPU (weak scaling: every GPU owns a
disjoint range of realisations); with N > 1 every step ends with the all-gather of the per-iteration powers and the
all-reduce of the dB histogram.  Inputs (spectrum, pupil weights) are resident in HBM before the timed region; only the
80 kB of results per step and GPU crosses PCIe.  Rank 0 prints ONE JSON line.

`--workload config3` is BASELINE.json configs[3] instead: 2048 x 2048 grid, 100 000 iterations per step IN TOTAL, cut into
N equal contiguous ranges (strong scaling), same exchange.

With N > 1 a step synchronises each device once: the kernels are enqueued without waiting, the RCCL collectives follow on
the same streams, only the gathered result is copied back.  Every exchange runs under a deadline (FASTMC_EXCHANGE_TIMEOUT,
default 120 s): a collective that does not come back is aborted (ncclCommAbort) and the run goes on with the host exchange
-- the line then says so in `config.result_exchange` -- instead of hanging.  `config.rccl_ranks` is the world size the
communicator itself reports, `exchange` the time the collectives took (HIP events on the streams: transfer + waiting for
the slowest peer) and `pipeline.gpu_busy_ms_per_step` the min / max over the workers of their kernels' time.

Every figure in the line is computed from this run, from the instruction counts of the code object that ran
(fast_amd/kernel_isa_stats.json, written by the build) or, where it needs hardware counters, from a committed
rocprofv3 summary that is named in the line and dropped when it belongs to another build.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
F32_VECTOR_PEAK_TFLOPS = 157.3
NOMINAL_GHZ = 2.4
N_SIMD = 1024
ITERS_PER_STEP = 10000          # per GPU and step, workload configs[1] (set by main for configs[3])
HIST = (-60.0, 10.0, 4096)
# Issue cost per wave-instruction per SIMD in cycles at 2.4 GHz, measured by tools/ubench on MI355X at 4 waves per SIMD
# (profiles/r01j_ubench_valu_issue_rates.txt): v_add/mul/fma_f64 5.0, v_cvt_f64_f32 4.5, v_log/sqrt/sin/cos_f32 7.8
# (= 2 x 5.15 - 2.5 of the paired add), v_mad_u64_u32 8.1, plain integer / float32 2.7.
ISSUE_COST = {"valu_f64": 5.0, "valu_cvt_f64": 4.5, "valu_trans": 7.8, "valu_int_quarter": 8.1, "valu_other": 2.7}
ISSUE_COST_SOURCE = "profiles/r01j_ubench_valu_issue_rates.txt"


def workload_params(args):
    import fast_amd
    h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
    return {
        "NPXLS": args.npxls, "DX": 0.01, "NITER": 10000, "NCHUNKS": 100, "TEMPORAL": False,
        "SUBHARM": False, "SEED": 1, "LOGLEVEL": "ERROR", "W0": "opt", "D_GROUND": 0.8, "OBSC_GROUND": 0,
        "D_SAT": 0.1, "H_SAT": 36e6, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w,
        "WIND_DIR": np.array([0., 90., 180., 270.]), "L0": np.inf, "l0": 1e-6, "ZENITH_ANGLE": 55,
        "DTHETA": [4, 0], "AO_MODE": args.ao_mode, "DSUBAP": 0.1, "TLOOP": 1e-3, "TEXP": 1e-3, "ALIAS": True,
        "NOISE": 0, "GPU_PRECISION": args.precision, "GPU_RNG": "device", "GPU_RNG_PRECISION": getattr(args, "rng_precision", "f64"),
        "FFTW": True, "GPU_SHARD": False,
    }


def _cpu_info():
    model, phys = "unknown", None
    try:
        cores = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                pid = cid = None
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys


def cpu_baseline(sim, seconds_target=12.0):
    """The oracle (numpy restatement of the reference's CPU path, FFTW-branch semantics) timed on this host on a bounded
    sample of the same workload: one core (the reference's default, FFTW_THREADS 1, fast/conf.py:71-72) and, as an extra,
    one process on every CPU this process may run on (os.sched_getaffinity)."""
    from oracle import fastref as R
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, "1")
    ps, df, W, dx, lv = sim.powerspec, sim._prob.df, sim._prob.W, sim.dx, float(sim.logamp_var)
    t0 = time.perf_counter()
    R.monte_carlo(123, 20, 1, ps, df, W, dx, lv)          # warm-up chunk: 20 iterations
    t_probe = time.perf_counter() - t0
    chunks = int(max(1, min(10, seconds_target / 3 / max(t_probe, 1e-3))))
    n_it = 20 * chunks
    rates = []
    for rep in range(3):                                   # three repeats, median
        t0 = time.perf_counter()
        r = R.monte_carlo(124 + rep, n_it, chunks, ps, df, W, dx, lv)
        rates.append(n_it / (time.perf_counter() - t0))
        assert np.isfinite(r).all()
    model, phys = _cpu_info()
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {"value": float(np.median(rates)), "unit": "iterations/s", "cores": 1, "kind": "port",
           "sample": f"median of 3 x {n_it} iterations ({chunks} chunks of 20) of the same {ps.shape[0]}^2 workload, "
                     f"oracle/fastref.py (numpy {np.__version__} pocketfft, float64, 1 thread); "
                     f"repeats {', '.join(f'{x:.1f}' for x in rates)} it/s",
           "host_cpus": os.cpu_count(), "usable_cpus": affinity, "physical_cores": phys, "cpu_model": model}
    try:   # every usable CPU: one child program each (never a fork of this GPU-initialised process)
        import subprocess
        import tempfile
        ncore = affinity
        try:       # each child holds ~0.5 GB at 1024^2 (20 complex128 screens + copies): stay inside the free memory
            avail_kb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
            ncore = max(1, min(ncore, int(avail_kb / 1024 / 1024 / 0.7)))
        except Exception:
            pass
        with tempfile.TemporaryDirectory() as tmp:
            npz = os.path.join(tmp, "inputs.npz")
            np.savez(npz, ps=ps, df=df, W=W, dx=dx, lv=lv)
            t0 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", npz, str(1000 + i), "1"], cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(ncore)]
            outs = [p.communicate(timeout=900)[0] for p in procs]
            wall = time.perf_counter() - t0
        if all(p.returncode == 0 for p in procs):
            n_done = sum(int(o.split()[0]) for o in outs)
            t_comp = max(float(o.split()[1]) for o in outs)          # slowest child's compute loop (no interpreter start)
            out["all_cores"] = {"value": n_done / t_comp, "unit": "iterations/s", "cores": ncore,
                                "sample": f"{ncore} concurrent child processes (one per usable CPU) x 20 iterations; slowest compute "
                                          f"loop {t_comp:.1f} s, wall incl. interpreter start {wall:.1f} s"}
        else:
            out["all_cores"] = {"error": "a worker failed"}
    except Exception as e:   # the baseline of record is the single-core figure above
        out["all_cores"] = {"error": str(e)}
    return out


def extras(args, device):
    """Side measurements printed next to the headline (never part of `value`): the float32 pipeline on the same job, the
    same job on a round decimal grid (NPXLS 1000: the 50-lane kernel family),
    BASELINE configs[2] (AO-corrected residual spectrum) with the warm time of the power-spectrum kernel that config adds
    to every `Fast()`, configs[3] (2048^2, 100 000 iterations, split over two handles = two worker threads, here on one
    device) and configs[4] (32 zenith angles x 4096 iterations at 1024^2, AO) with the init / Monte-Carlo wall split."""
    import copy
    import fast_amd
    from fast_amd import sweep
    out = {}
    for tag, over in (("f32_same_job", {"GPU_PRECISION": "f32", "GPU_RNG_PRECISION": "f32"}), ("config2_AO_alias_f64", {"AO_MODE": "AO", "ALIAS": True}),
                      ("npxls1000_f64_lanes50_kernels", {"NPXLS": 1000})):
        p = workload_params(copy.copy(args))
        p.update(over)
        p["GPU_DEVICE"] = device
        t0 = time.perf_counter()
        sim = fast_amd.Fast(p)
        init_s = time.perf_counter() - t0
        h = sim._handle
        n_real = 10000 // 2
        h.run(1, 0, n_real, None, float(sim.logamp_var), False)
        t0 = time.perf_counter()
        for i in range(3):
            h.run(1, (i + 1) * n_real, n_real, None, float(sim.logamp_var), False)
        dt = time.perf_counter() - t0
        sim.compute_powerspec()                              # second evaluation: no module load in the figure
        out[tag] = {"iterations_per_s": 3 * 10000 / dt, "init_s": init_s, "powerspec_kernel_ms_warm": sim.powerspec_kernel_ms,
                    "mean_dB_rel": float(10 * np.log10(np.mean(h.run(1, 0, 512, None, float(sim.logamp_var), False))))}
    # configs[3]: 2048^2, 100 000 iterations, two handles
    p = workload_params(copy.copy(args))
    p.update({"NPXLS": 2048, "NITER": 100000, "NCHUNKS": 100, "GPU_DEVICES": [device, device]})
    t0 = time.perf_counter()
    sim = fast_amd.Fast(p)
    t1 = time.perf_counter()
    r = sim.run()._r
    t2 = time.perf_counter()
    hist = sim.histogram(*HIST)
    out["config3_2048_100k_two_handles"] = {"iterations_per_s": 100000 / (t2 - t1), "init_s": t1 - t0, "run_s": t2 - t1,
                                            "histogram_total": int(hist.sum()), "exchange": sim._group.exchange,
                                            "mean_dB_rel": float(10 * np.log10(r.mean()))}
    # configs[4]: zenith-angle scan, 32 x 4096 iterations at 1024^2, AO + alias (second pass: pupil / module caches warm)
    base = workload_params(copy.copy(args))
    base.update({"AO_MODE": "AO", "ALIAS": True, "GPU_DEVICE": device, "NPXLS": 1024})
    angles = np.linspace(0, 70, 32)
    sweep.zenith_scan(base, angles[:2], niter=4096)
    t0 = time.perf_counter()
    recs = sweep.zenith_scan(base, angles, niter=4096)
    wall = time.perf_counter() - t0
    out["config5_zenith_scan_32x4096"] = {"wall_s": wall, "init_s": sum(r["init_s"] for r in recs), "run_s": sum(r["run_s"] for r in recs),
                                          "iterations_per_s_end_to_end": 32 * 4096 / wall,
                                          "powerspec_kernel_ms_total": sum(r["powerspec_kernel_ms"] for r in recs),
                                          "mean_dB_rel_first_last": [recs[0]["mean_dB_rel"], recs[-1]["mean_dB_rel"]]}
    out.update(extras_unmeasured_rows(args, device))
    return out


def extras_multi(args, mode, devices, rank, world, rdzv):
    """N > 1, EVERY rank calls this (its exchanges are collective): BASELINE configs[3] (2048^2, 100 000 iterations in total, sharded
    over ALL the GPUs of the launch: strong scaling, gather + dB histogram) and configs[4] (32 zenith angles x 4096 iterations at
    1024^2, AO + alias, dealt 32 / N per GPU), in the launch form of the headline: one process with a worker thread per device
    (`GPU_DEVICES`, `sweep.zenith_scan(devices=...)`) or one process per GPU (`GPU_SHARD` auto, `zenith_scan(rank, world)` +
    `gather_records`).  The reference's pattern: fast/complete_orbit_simulation.py:217-228 (one Fast object per geometry sample)."""
    import copy
    import fast_amd
    from fast_amd import sweep
    workers = len(devices) if mode == "threads" else world
    out = {}

    def wall_max(dt):
        return dt if rdzv is None else float(rdzv.all_reduce(np.array([dt]), "max")[0])

    # configs[3]
    p = workload_params(copy.copy(args))
    p.update({"NPXLS": 2048, "NITER": 100000, "NCHUNKS": 100, "AO_MODE": "NOAO"})
    if mode == "threads":
        p["GPU_DEVICES"] = devices
    else:
        p["GPU_DEVICE"] = devices[0]
        p["GPU_SHARD"] = True                                # one process per GPU: Fast.run shards over the ranks and exchanges once
    if 50000 % workers == 0:
        t0 = time.perf_counter()
        sim = fast_amd.Fast(p)
        sim.run()                                            # first run: allocations, module load of the 2048 kernels
        t1 = time.perf_counter()
        if rdzv is not None:
            rdzv.barrier()
        t2 = time.perf_counter()
        r = sim.run()._r
        run_s = wall_max(time.perf_counter() - t2)
        hist = sim.histogram(*HIST)
        if mode == "threads":
            exch, ranks = sim._group.exchange, (sim._group.rccl_ranks if sim._group.exchange == "rccl" else 0)
        else:
            tr = sim._transport()
            exch, ranks = tr.name, (tr.rccl_ranks if tr.name == "rccl" else 0)
        out["config3_2048_100k_all_gpus"] = {"iterations_per_s": 100000 / run_s, "run_s": run_s, "init_and_first_run_s": t1 - t0, "workers": workers,
                                             "scaling": "strong", "histogram_total": int(hist.sum()), "result_exchange": exch, "rccl_ranks": int(ranks),
                                             "mean_dB_rel": float(10 * np.log10(r.mean()))}
        del sim
    else:
        out["config3_2048_100k_all_gpus"] = {"skipped": f"{workers} workers do not divide 50 000 realisations"}
    # configs[4]
    base = workload_params(copy.copy(args))
    base.update({"AO_MODE": "AO", "ALIAS": True, "NPXLS": 1024, "GPU_DEVICE": devices[0]})
    angles = np.linspace(0, 70, 32)
    devs = devices if mode == "threads" else None
    r_, w_ = (0, 1) if mode == "threads" else (rank, world)
    sweep.zenith_scan(base, angles[:2 * workers], niter=4096, rank=r_, world=w_, devices=devs)      # warm: pupil / module caches, handle caches
    if rdzv is not None:
        rdzv.barrier()
    t0 = time.perf_counter()
    recs = sweep.zenith_scan(base, angles, niter=4096, rank=r_, world=w_, devices=devs)
    if mode == "ranks":
        recs = sweep.gather_records(recs)
    wall = wall_max(time.perf_counter() - t0)
    per_dev = {}
    for rec in recs:
        per_dev[rec["device"]] = per_dev.get(rec["device"], 0) + 1
    out["config5_zenith_scan_32x4096_all_gpus"] = {
        "wall_s": wall, "iterations_per_s_end_to_end": 32 * 4096 / wall, "samples": len(recs), "workers": workers,
        "samples_per_worker": 32 / workers, "samples_by_device_index": {str(k): v for k, v in sorted(per_dev.items())},
        "init_s_sum": sum(r["init_s"] for r in recs), "run_s_sum": sum(r["run_s"] for r in recs),
        "mean_dB_rel_first_last": [recs[0]["mean_dB_rel"], recs[-1]["mean_dB_rel"]],
        "note": "no result exchange but the per-sample records (JSON over the rendezvous between processes); device indices are per process"}
    return out


def extras_unmeasured_rows(args, device):
    """The rows of SURVEY section 8 that the headline does not exercise, each timed here so that they are on the record:
    BASELINE configs[0] exactly as stated (the reference's shipped test/test_params.py: NPXLS 256, TEMPORAL on, 100 iterations)
    on the GPU and on the CPU oracle; a TEMPORAL run at 1024^2; SUBHARM on; 128^2, 256^2 and 512^2 device-mode rates (packed rows); and the
    opt-in float32 draw (GPU_RNG_PRECISION 'f32': what the shortcut buys over the default, the reference's float64 precision)."""
    import copy
    import fast_amd
    from oracle import fastref as R
    out = {}

    def shipped(**over):
        p = workload_params(copy.copy(args))
        p.update({"NPXLS": 256, "NITER": 100, "NCHUNKS": 10, "TEMPORAL": True, "DT": 0.001, "AO_MODE": "AO", "ALIAS": True, "FFTW": True,
                  "GPU_DEVICE": device, "GPU_PRECISION": "f64"})
        p.update(over)
        return p

    # configs[0]: GPU (second object and second run: module load and allocations out of the figure) and the CPU oracle
    fast_amd.Fast(shipped()).run()
    t0 = time.perf_counter()
    sim = fast_amd.Fast(shipped())
    t1 = time.perf_counter()
    r = sim.run()._r
    t2 = time.perf_counter()
    W, prob = sim._prob.W, sim._prob
    tc = time.perf_counter()
    r_cpu = R.monte_carlo_temporal(1, 100, 10, sim.powerspec_per_layer, prob.df, W, sim.dx, float(sim.logamp_var), sim.temporal_logamp_powerspec,
                                   sim.wind_vector, 0.001, sim.Npxls, sim.Npxls_pup)
    t_cpu = time.perf_counter() - tc
    out["config0_shipped_example_256_temporal_100it"] = {
        "gpu_init_s": t1 - t0, "gpu_run_s": t2 - t1, "gpu_iterations_per_s": 100 / (t2 - t1), "cpu_oracle_run_s": t_cpu,
        "cpu_oracle_iterations_per_s": 100 / t_cpu, "max_rel_diff_gpu_vs_cpu_oracle": float(np.abs(r / r_cpu - 1).max()),
        "note": "same SEED: the GPU run reproduces the oracle's (= the reference's) series; the CPU figure excludes compute_powerspec"}
    # TEMPORAL at the BASELINE grid: 1024^2, 2000 time steps
    p = shipped(NPXLS=1024, NITER=2000, NCHUNKS=10)
    fast_amd.Fast(copy.copy(p)).run()
    t0 = time.perf_counter()
    sim = fast_amd.Fast(copy.copy(p))
    t1 = time.perf_counter()
    sim.run()
    t2 = time.perf_counter()
    out["temporal_1024_2000_steps"] = {"init_s": t1 - t0, "run_s": t2 - t1, "iterations_per_s": 2000 / (t2 - t1)}
    # device-mode rates: sub-harmonics on, small grids, float64 generator
    for tag, over in (("subharm_on_1024_f64", {"SUBHARM": True, "L0": 25.0}), ("npxls128_f64", {"NPXLS": 128}), ("npxls256_f64", {"NPXLS": 256}), ("npxls512_f64", {"NPXLS": 512}),
                      ("f32_draw_1024_f64", {"GPU_RNG_PRECISION": "f32"})):
        p = workload_params(copy.copy(args))
        p.update(over)
        p["GPU_DEVICE"] = device
        sim = fast_amd.Fast(p)
        h = sim._handle
        n_it = 40000 if p["NPXLS"] <= 512 else 10000
        h.run(1, 0, n_it // 2, None, float(sim.logamp_var), False)
        t0 = time.perf_counter()
        for i in range(3):
            h.run(1, (i + 1) * (n_it // 2), n_it // 2, None, float(sim.logamp_var), False)
        out[tag] = {"iterations_per_s": 3 * n_it / (time.perf_counter() - t0), "rows_kernel": h.last_kernels()[0]}
    # same-seed modes (the reference's numbers for its SEED): numpy's stream drawn on the device (GPU_RNG 'numpy') against numpy's
    # draws on the host (GPU_RNG 'host'), same chunking as BASELINE configs[1] (100 iterations per chunk), results compared
    p = workload_params(copy.copy(args))
    p.update({"GPU_DEVICE": device, "NITER": 200, "NCHUNKS": 2, "GPU_RNG": "host"})
    t0 = time.perf_counter()
    r_host = fast_amd.Fast(copy.copy(p)).run()._r
    t_host = time.perf_counter() - t0
    p["GPU_RNG"] = "numpy"
    r_dev = fast_amd.Fast(copy.copy(p)).run()._r                       # (also warms the tables and buffers)
    p.update({"NITER": 20000, "NCHUNKS": 200})
    t_runs = []
    for _ in range(3):                                                 # (a run is ~0.2 s of chunks of 100: the median of three)
        sim = fast_amd.Fast(copy.copy(p))
        t0 = time.perf_counter()
        sim.run()
        t_runs.append(time.perf_counter() - t0)
    t_dev = sorted(t_runs)[1]
    out["same_seed_1024"] = {"host_draws_iterations_per_s": 200 / t_host, "device_numpy_stream_iterations_per_s": 20000 / t_dev,
                             "device_numpy_stream_runs": [20000 / t for t in t_runs],
                             "ratio": (20000 / t_dev) / (200 / t_host), "max_rel_diff_same_seed": float(np.abs(r_dev / r_host - 1).max()),
                             "note": "GPU_RNG 'numpy': numpy's PCG64 + ziggurat stream reproduced on the device (fast_amd/csrc/fmc_npstream.h); "
                                     "20 000 iterations in chunks of 100, median of three runs; the host figure includes Fast() construction of a 200-iteration run"}
    return out


def load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def roofline(args, N, Np, tim, steps, workers, iters_per_worker_step, kernels=("", "")):
    """The `roofline` object for the dominant kernel (the row kernel) of THIS run.  The kernel is bound by the SIMDs'
    instruction issue (float64 butterflies + the generator + LDS instructions), not by HBM or MFMA:
    `bound` says so, `achieved` / `frac` are executed float64 (float32) vector FLOP/s against the vector peak, `issue` the
    share of the SIMDs' issue cycles the VALU instructions of the code object account for, `hbm` the byte models.
    `kernels` = (rows, cols) names the library reports for what it launched (fastmc_last_kernels): the instruction counts are
    those of THAT instantiation (fast_amd/kernel_isa_stats.json; split rows of 2048 / 4096: row loop + S passes of the sub-row loop)."""
    f64 = kernels[0].split("<")[-1].startswith("double") if kernels[0] else args.precision == "f64"     # what ran, not what was asked for
    launches = max(tim["rows_launches"], 1)
    avg_rows_ms = tim["rows_ms"] / launches
    avg_cols_ms = tim["cols_ms"] / max(tim["cols_launches"], 1)
    real_per_launch = iters_per_worker_step / 2 * steps * workers / launches        # realisations in an average launch
    isa = load_json(os.path.join(ROOT, "fast_amd", "kernel_isa_stats.json")) or {}
    st = next((v for v in isa.values() if isinstance(v, dict) and v.get("kernel") == kernels[0]), None)
    wc = 16 if f64 else 8
    it_per_launch = 2 * real_per_launch
    bytes_alg = (20 if f64 else 10) * N * N                                     # SURVEY 8(d), per iteration
    # what THIS design must move per realisation: V written by the row pass and read by the column pass (window columns
    # only), the column partials, the results; the float32 colouring table (4 N^2 B) stays in cache across a launch
    bytes_pruned_rows = wc * N * Np
    bytes_pruned_cols = wc * N * Np + 32 * Np + 8 * Np * Np / max(real_per_launch, 1)
    out = {"bound": "valu", "kernel": kernels[0] or "unknown", "cols_kernel": kernels[1], "avg_launch_ms": avg_rows_ms, "realisations_per_launch": real_per_launch,
           "iterations_per_launch": it_per_launch, "peak": F64_VECTOR_PEAK_TFLOPS if f64 else F32_VECTOR_PEAK_TFLOPS,
           "unit": "TFLOP/s", "achieved": None, "frac": None, "traffic": None}
    if st:
        rows = real_per_launch * N                # row iterations of the kernel's main loop per launch
        flop_lane = st["f64_flop_per_lane"] if f64 else st["f32_flop_per_lane"]
        flops = flop_lane * 64 * rows
        out["achieved"] = flops / (avg_rows_ms * 1e-3) / 1e12
        out["frac"] = out["achieved"] / out["peak"]
        cyc = sum(ISSUE_COST.get(k, 0.0) * v for k, v in st["instructions"].items())
        row_cycles = avg_rows_ms * 1e-3 * NOMINAL_GHZ * 1e9 * N_SIMD / rows
        out["issue"] = {"frac": cyc / row_cycles, "valu_issue_cycles_per_row": cyc, "measured_cycles_per_row_at_2.4GHz": row_cycles,
                        "instructions_per_row": st["instructions"], "valu_instructions_per_row": st["valu_total"],
                        "flop_per_lane_per_row": flop_lane, "counts_from": "fast_amd/kernel_isa_stats.json (tools/isa_stats.py on the built source)",
                        "cycles_per_instruction": ISSUE_COST, "costs_from": ISSUE_COST_SOURCE,
                        "note": "VALU classes only; LDS / scalar / memory instructions issue on top of this"}
    hbm = {"rows": {"algorithmic_GBps": bytes_alg * it_per_launch / (avg_rows_ms * 1e-3) / 1e9,
                    "pruned_algorithmic_GBps": bytes_pruned_rows * real_per_launch / (avg_rows_ms * 1e-3) / 1e9},
           "cols": {"pruned_algorithmic_GBps": bytes_pruned_cols * real_per_launch / (avg_cols_ms * 1e-3) / 1e9 if avg_cols_ms else None},
           "peak_GBps": HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": bytes_alg,
           "pruned_bytes_per_realisation": {"rows": bytes_pruned_rows, "cols": bytes_pruned_cols},
           "note": "algorithmic = SURVEY 8(d) model (full grid written and re-read between the passes); the kernels are output-"
                   "pruned and move only the pruned bytes, so the 8(d) figure is not a bandwidth and may exceed the peak"}
    hbm["rows"]["frac_pruned"] = hbm["rows"]["pruned_algorithmic_GBps"] / HBM_PEAK_GBS
    if hbm["cols"]["pruned_algorithmic_GBps"]:
        hbm["cols"]["frac_pruned"] = hbm["cols"]["pruned_algorithmic_GBps"] / HBM_PEAK_GBS
    # hardware counters of the same command, when a committed rocprofv3 summary belongs to this build
    # (profiles/latest_counters*.json: one file per profiled row kernel -- headline, float64 generator, 2048^2 ...)
    import glob
    profs = [load_json(f) for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "latest_counters*.json")))]
    prof = next((q for q in profs if q and q.get("rows_kernel") == kernels[0] and q.get("npxls") == N), None)
    if prof and st and prof.get("precision") == args.precision:
        if prof.get("rows_valu_instructions_per_row") == st["valu_total"]:
            for k in ("rows", "cols"):
                c = prof.get(k, {})
                if c.get("hbm_bytes_per_launch") and c.get("avg_launch_ms"):
                    hbm[k]["counter_GBps"] = c["hbm_bytes_per_launch"] / (c["avg_launch_ms"] * 1e-3) / 1e9
                    hbm[k]["frac_counter"] = hbm[k]["counter_GBps"] / HBM_PEAK_GBS
            tb, rpl = prof.get("rows", {}).get("hbm_bytes_per_launch"), prof.get("realisations_per_launch")
            # counter bytes of the profiled launches, scaled to this run's launch size (bytes are per realisation)
            out["traffic"] = tb * real_per_launch / rpl if (tb and rpl) else tb
            out["counters"] = {k: prof[k] for k in ("valu_busy", "issue_busy", "lds_issue_busy", "valu_busy_counter_ratio", "issue_busy_counter_ratio",
                                                    "lds_issue_busy_counter_ratio", "busy_note", "clock_GHz_profiled", "source") if k in prof}
        else:
            out["counters"] = {"stale": f"{prof.get('source')} belongs to a build with {prof.get('rows_valu_instructions_per_row')} "
                                        f"VALU instructions per row, this build has {st['valu_total']}"}
    out["hbm"] = hbm
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--rng-precision", default="f64", choices=["f32", "f64"],
                    help="device generator of the TIMED steps: f64 (default, the headline: the reference's 53-bit normals and float64 "
                         "colouring, fast/funcs.py:352-356, fast/fast.py:593-594) or f32 (the opt-in shortcut GPU_RNG_PRECISION 'f32': "
                         "float32 normals and colouring); without this flag the same job is timed a second time with the f32 draw and "
                         "reported as the extra value_f32_draw")
    ap.add_argument("--npxls", type=int, default=None, help="grid size (default: 1024, or 2048 with --workload config3)")
    ap.add_argument("--ao-mode", default="NOAO")
    ap.add_argument("--workload", default="config1", choices=["config1", "config3"],
                    help="config1 = BASELINE configs[1]: 1024^2, 10 000 iterations per step and GPU (weak scaling, the headline); "
                         "config3 = BASELINE configs[3]: 2048^2, 100 000 iterations per step in total, split over the GPUs (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (f32, AO config, configs[3], configs[4])")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 5 s sustained run after the timed steps")
    ap.add_argument("--iters-per-step", type=int, default=ITERS_PER_STEP, help="iterations per GPU and step of the configs[1] workload "
                    "(default 10000 = BASELINE configs[1]; other values are for overhead studies, the line says what ran)")
    ap.add_argument("--no-host-cost-pass", action="store_true", help="skip the one-call reference run behind pipeline.host_ms_per_step")
    ap.add_argument("--no-pipeline", action="store_true", help="one step at a time (enqueue, exchange, wait) instead of two steps in flight per device")
    ap.add_argument("--no-f32-draw-pass", "--no-f64-generator-pass", dest="no_other_precision_pass", action="store_true",
                    help="skip the second timed pass with the other generator precision")
    ap.add_argument("--batch", type=int, default=0, help="realisations per launch (0 = library default)")
    ap.add_argument("--require-rccl", action="store_true",
                    help="with N > 1: exit non-zero unless every timed step's exchange ran over RCCL with a communicator of N ranks "
                         "(a host fall-back must not pass for a scaling point)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    strong = args.workload == "config3"
    if args.npxls is None:
        args.npxls = 2048 if strong else 1024

    import fast_amd
    from fast_amd import _lib, dist, multi, rendezvous
    rank, world, local_rank = rendezvous.env_world()
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("no GPU visible: bench.py measures the HIP path and has no CPU fallback")
    rdzv = None
    if world > 1:
        # one process per GPU, started by a launcher
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        mode = "ranks"
        devices = [int(os.environ.get("FASTMC_BENCH_DEVICE", -1))]
        if devices[0] < 0:
            devices = [_lib.default_device()]              # raises when LOCAL_RANK has no GPU of its own
        rdzv = rendezvous.from_env()
    else:
        # one process drives all the GPUs; FASTMC_BENCH_DEVICES=0,0 puts several workers on one device (functional tests)
        mode = "threads" if args.gpus > 1 else "single"
        if os.environ.get("FASTMC_BENCH_DEVICES"):
            devices = [int(x) for x in os.environ["FASTMC_BENCH_DEVICES"].split(",")]
            if len(devices) != args.gpus:
                raise SystemExit(f"FASTMC_BENCH_DEVICES names {len(devices)} workers, --gpus {args.gpus}")
        else:
            if args.gpus > ndev:
                raise SystemExit(f"--gpus {args.gpus} but only {ndev} GPU(s) visible to this process "
                                 "(under a launcher, start one rank per GPU; bench.py never reports GPUs that did not run)")
            devices = list(range(args.gpus))
    if max(devices) >= ndev:
        raise SystemExit(f"device {max(devices)} requested but only {ndev} GPU(s) visible")
    workers = len(devices) if mode != "ranks" else world
    if strong:
        if 50000 % workers:
            raise SystemExit("--workload config3 splits 50 000 realisations: the number of GPUs must divide it")
        iters_worker = 100000 // workers                  # iterations per worker and step (strong scaling)
    else:
        iters_worker = args.iters_per_step                 # weak scaling: fixed work per GPU
    n_real = iters_worker // 2

    p = workload_params(args)
    p["GPU_DEVICES"] = devices
    p["GPU_BATCH"] = args.batch
    t0 = time.perf_counter()
    sim = fast_amd.Fast(p)
    init_s = time.perf_counter() - t0
    grp, h = sim._group, sim._handle
    N, Np = sim.Npxls, sim.Npxls_pup
    lvar = float(sim.logamp_var)

    tr = None
    if mode == "ranks":
        tr = dist.make_transport(h, rdzv)                 # collective: RCCL on every rank, or the host path on every rank

    def exchange_name():
        if mode == "ranks":
            return "rccl (in-library, one process per GPU)" if tr.name == "rccl" else f"host sockets ({getattr(tr, 'why', '')})"
        if mode == "threads":
            return "rccl (in-library, ncclCommInitAll)" if grp.exchange == "rccl" else grp.exchange
        return "none"

    hist_total = None
    acc = {"exchange_device_ms": [], "exchange_wall_ms": 0.0, "rccl_steps": 0, "host_steps": 0}

    def step(i, record=False):
        """Step i: every worker computes `iters_worker` iterations of its own realisation range, then one exchange."""
        nonlocal hist_total
        base = i * workers * n_real
        if mode == "ranks":
            out, hist_total, info = dist.step_sharded(h, tr, p["SEED"], base, workers * n_real, lvar, False, HIST)
            if record:
                acc["exchange_device_ms"].append([info["exchange_device_ms"]])
                acc["exchange_wall_ms"] += info.get("exchange_host_ms", 0.0)
                acc["rccl_steps" if info["exchange"] == "rccl" else "host_steps"] += 1
            return out
        out = grp.run(p["SEED"], base, workers * n_real, None, lvar, False, hist_range=HIST)
        hist_total = grp.last_hist
        if record and workers > 1:
            if grp.last_exchange == "rccl":
                acc["exchange_device_ms"].append(list(grp.last_exchange_ms))
                acc["rccl_steps"] += 1
            else:
                acc["exchange_wall_ms"] += grp.last_exchange_wall_ms
                acc["host_steps"] += 1
        return out

    def run_steps(first, count, tim=None, record=False):
        """Steps first ... first + count - 1 with TWO steps in flight per device (fast_amd.multi.DeviceGroup.run_pipelined /
        fast_amd.dist.steps_pipelined: step i + 1 is enqueued before step i's results are waited for); --no-pipeline runs them
        one after the other as rounds 1-3 did.  Same realisation ranges, same results either way."""
        nonlocal hist_total
        out = None
        if args.no_pipeline:
            for i in range(count):
                out = step(first + i, record=record)
                if tim is not None:
                    add_timing(tim)
            return out
        spans = [((first + i) * workers * n_real, workers * n_real) for i in range(count)]
        if mode == "ranks":
            for out, hist_total, info in dist.steps_pipelined(h, tr, p["SEED"], spans, lvar, False, HIST):
                if record:
                    acc["exchange_device_ms"].append([info["exchange_device_ms"]])
                    acc["exchange_wall_ms"] += info.get("exchange_host_ms", 0.0)
                    acc["rccl_steps" if info["exchange"] == "rccl" else "host_steps"] += 1
                if tim is not None:
                    add_timing(tim)
            return out
        for out, hist_total in grp.run_pipelined(p["SEED"], spans, lvar, False, HIST):
            if record and workers > 1:
                if grp.last_exchange == "rccl":
                    acc["exchange_device_ms"].append(list(grp.last_exchange_ms))
                    acc["rccl_steps"] += 1
                else:
                    acc["exchange_wall_ms"] += grp.last_exchange_wall_ms
                    acc["host_steps"] += 1
            if tim is not None:
                add_timing(tim)
        return out

    def sync_all():
        if rdzv is not None:
            rdzv.barrier()                  # library calls are blocking: every device is idle when its rank gets here

    tim_keys = ("rows_ms", "cols_ms", "finalize_ms", "rows_launches", "cols_launches")
    busy = []                               # per step: per worker kernels' time (rows + cols + finalize, HIP events)

    def add_timing(tim):
        per = []
        for t in (grp.last_timing() if mode != "ranks" else [h.last_timing()]):
            for k in tim_keys:
                tim[k] += t[k]
            per.append(t["rows_ms"] + t["cols_ms"] + t["finalize_ms"])
        busy.append(per)

    run_steps(0, args.warmup)
    tim = dict.fromkeys(tim_keys, 0.0)
    sync_all()
    t0 = time.perf_counter()
    out = run_steps(args.warmup, args.steps, tim, record=True)
    sync_all()
    dt = time.perf_counter() - t0
    kernels = h.last_kernels()                                            # what the timed steps launched
    # effective shader clock INSIDE the last row launch of the timed steps (fastmc_last_clock: shader-clock ticks per constant-rate
    # tick over one workgroup's life in the middle of the launch), per local worker
    clocks = [c for c in (grp.each(lambda hh, i: hh.last_clock()) if mode != "ranks" else [h.last_clock()]) if c]
    busy = np.asarray(busy, dtype=float).reshape(args.steps, -1)          # (steps, local workers)
    ex_dev = np.asarray(acc["exchange_device_ms"], dtype=float)
    ex_dev = ex_dev.reshape(len(acc["exchange_device_ms"]), -1) if ex_dev.size else np.zeros((0, 1))
    if rdzv is not None:
        dt = float(rdzv.all_reduce(np.array([dt]), "max")[0])
        for k in tim_keys:
            tim[k] = float(rdzv.all_reduce(np.array([tim[k]]), "sum")[0])
        busy = np.concatenate(list(rdzv.all_gather_array(busy)), axis=1)   # (steps, world)
        n_ex = rdzv.all_gather_array(np.array([ex_dev.shape[0]], dtype=np.int64)).ravel()
        if ex_dev.size and len(set(n_ex.tolist())) == 1:
            ex_dev = np.concatenate(list(rdzv.all_gather_array(ex_dev)), axis=1)
    assert np.isfinite(out).all() and (out > 0).all()

    sustained = None
    if not args.no_sustained:
        # one figure a coarse sampler (rocm-smi at 1 Hz) can see: the same steps back to back for >= 5 s
        sync_all()
        t0 = time.perf_counter()
        n_sus = 0
        while True:
            run_steps(args.warmup + args.steps + n_sus, 16)
            n_sus += 16
            go = np.array([1 if time.perf_counter() - t0 < 5.0 else 0])
            if rdzv is not None:
                go = rdzv.all_reduce(go, "max")
            if not go[0]:
                break
        sync_all()
        dts = time.perf_counter() - t0
        if rdzv is not None:
            dts = float(rdzv.all_reduce(np.array([dts]), "max")[0])
        sustained = {"seconds": dts, "steps": n_sus, "value": iters_worker * n_sus * workers / dts, "unit": "iterations/s"}

    # What the host costs per step: the SAME realisations as the timed steps once more as ONE call per worker (args.steps times
    # the iterations, one exchange) -- the device-limited time of the job; the timed steps' wall time beyond it is what issuing,
    # exchanging and collecting K separate steps cost.  (The kernels' own event times cannot say it: with two steps in flight
    # the events of consecutive launches overlap.)
    host_cost = None
    if not args.no_host_cost_pass:
        keep_hist = hist_total
        sync_all()
        t0 = time.perf_counter()
        if mode == "ranks":
            dist.step_sharded(h, tr, p["SEED"], args.warmup * workers * n_real, args.steps * workers * n_real, lvar, False, HIST)
        else:
            grp.run(p["SEED"], args.warmup * workers * n_real, args.steps * workers * n_real, None, lvar, False, hist_range=HIST)
        sync_all()
        dt_one = time.perf_counter() - t0
        # this pass's HIP events time the same launches with ONE run in the queue: with two steps in flight the events of
        # consecutive launches overlap by a few per cent (the start marker of a launch is stamped while the previous step still
        # drains), so the roofline's launch durations are taken from here
        tim_one = dict.fromkeys(tim_keys, 0.0)
        busy_keep2, busy = busy, []
        add_timing(tim_one)
        busy = busy_keep2
        if rdzv is not None:
            dt_one = float(rdzv.all_reduce(np.array([dt_one]), "max")[0])
            for k in tim_keys:
                tim_one[k] = float(rdzv.all_reduce(np.array([tim_one[k]]), "sum")[0])
        host_cost = {"one_call_ms_per_step": dt_one / args.steps * 1e3, "host_ms_per_step": (dt - dt_one) / args.steps * 1e3, "tim": tim_one}
        hist_total = keep_hist

    # The same job with the OTHER generator precision, timed like the headline: same steps, same barriers, fresh realisation
    # ranges.  The headline draws at the REFERENCE's precision (53-bit normals, float64 colouring: fast/funcs.py:352-356,
    # fast/fast.py:593-594); the extra pass is the opt-in float32 draw (`value_f32_draw`).  With `--rng-precision f32` the roles
    # swap (`value_f64_generator`).
    other = None
    other_prec = "f32" if args.rng_precision == "f64" else "f64"
    if args.precision == "f64" and not args.no_other_precision_pass:
        grp.each(lambda hh, i: hh.set_rng_precision(other_prec))
        first = args.warmup + args.steps + 100000
        run_steps(first, 1)
        busy_keep, busy = busy, []
        tim2 = dict.fromkeys(tim_keys, 0.0)
        sync_all()
        t0 = time.perf_counter()
        out2 = run_steps(first + 1, args.steps, tim2)
        sync_all()
        dt2 = time.perf_counter() - t0
        kernels2 = h.last_kernels()
        # launch durations for this pass's roofline from ONE more step issued on its own (see the host-cost pass above)
        tim2_one = dict.fromkeys(tim_keys, 0.0)
        step(first + 1 + args.steps)
        add_timing(tim2_one)
        busy = busy_keep
        if rdzv is not None:
            dt2 = float(rdzv.all_reduce(np.array([dt2]), "max")[0])
            for k in tim_keys:
                tim2[k] = float(rdzv.all_reduce(np.array([tim2[k]]), "sum")[0])
                tim2_one[k] = float(rdzv.all_reduce(np.array([tim2_one[k]]), "sum")[0])
        assert np.isfinite(out2).all() and (out2 > 0).all()
        other = {"dt": dt2, "tim": tim2, "tim_one": tim2_one, "kernels": kernels2}
        grp.each(lambda hh, i: hh.set_rng_precision(args.rng_precision))

    # GPUs that actually ran (one node: distinct device indices over all workers) and the communicator's own world size
    if mode == "ranks":
        n_devices = len(set(int(x) for x in rdzv.all_gather_array(np.array([devices[0]], dtype=np.int64)).ravel()))
        rccl_ranks = int(rdzv.all_reduce(np.array([tr.rccl_ranks if tr.name == "rccl" else 0]), "min")[0])
    else:
        n_devices = len(set(devices))
        rccl_ranks = grp.rccl_ranks if (mode == "threads" and grp.exchange == "rccl") else 0
    rccl_problem = None
    if args.require_rccl and workers > 1:
        if rccl_ranks != workers:
            rccl_problem = f"--require-rccl: the communicator reports {rccl_ranks} rank(s), {workers} wanted ({exchange_name()})"
        elif acc["host_steps"] or acc["rccl_steps"] != args.steps:
            rccl_problem = f"--require-rccl: {acc['rccl_steps']} of {args.steps} timed steps exchanged over RCCL, {acc['host_steps']} on the host ({exchange_name()})"
    # every row-kernel launch of worker 0 in issue order, as calls of n realisations (a call is n // batch launches of `batch` and
    # one of the remainder): tools/summarise_profile.py reads a profile's per-dispatch times against it
    B_launch = h.get_batch()
    calls = [["warmup", n_real, args.warmup], ["timed", n_real, args.steps]]
    if sustained:
        calls.append(["sustained", n_real, sustained["steps"]])
    if host_cost is not None:
        calls.append(["one_call", args.steps * n_real, 1])
    launch_plan = {"batch": B_launch, "kernel": kernels[0], "calls": calls, "launches_per_step": -(-n_real // B_launch),
                   "launches_of_one_step": [B_launch] * (n_real // B_launch) + ([n_real % B_launch] if n_real % B_launch else []),
                   "note": "realisations per call and worker; the other-precision pass and the extras launch other kernels"}
    clean = host_cost["tim"] if (host_cost is not None and not args.no_pipeline) else tim
    if rank == 0:
        total_iters = iters_worker * args.steps * workers
        value = total_iters / dt
        gpu_ms = tim["rows_ms"] + tim["cols_ms"] + tim["finalize_ms"]
        sim.compute_powerspec()                   # the evaluation at init paid the module load; this one is warm
        wl = (f"configs[3]: {N}^2 grid, Np={Np}, 100000 iters/step in total = {iters_worker} per GPU" if strong else
              f"configs[1]: {N}^2 grid, Np={Np}, {iters_worker} iters/step/GPU")
        exchange = {"steps_rccl": acc["rccl_steps"], "steps_host": acc["host_steps"]} if workers > 1 else None
        if exchange is not None and ex_dev.size:
            exchange["device_ms_per_step"] = {"mean": float(ex_dev.mean()), "min": float(ex_dev.min()), "max": float(ex_dev.max()),
                                              "note": "HIP events around the collectives on each worker's stream: transfer + the wait for the slowest peer"}
        if exchange is not None and acc["host_steps"]:
            exchange["host_ms_per_step"] = acc["exchange_wall_ms"] / acc["host_steps"]
        line = {
            "metric": f"Monte-Carlo iterations/sec ({N}^2 grid)", "value": value, "unit": "iterations/s",
            "n_gpus": n_devices, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": ("f64" if args.rng_precision == "f64" else "f64 (f32 draw)") if args.precision == "f64" else "f32", "data": "synthetic",
            "config": {"workload": wl + f", {args.ao_mode} von Karman spectrum, device generator (Philox4x32-7-seeded xoshiro128+ streams, Box-Muller)",
                       "arithmetic": ("complex128 transform and float64 detector sums" if args.precision == "f64" else "complex64 transform, float64 detector sums")
                                     + ("; the device generator draws 53-bit normals and colours in float64 like the reference (fast/funcs.py:352-356, "
                                        "fast/fast.py:593-594), fused into the row kernel" if args.rng_precision == "f64" else
                                        "; the device generator's normals are float32 (24-bit uniforms, hardware log/sqrt/sin/cos), "
                                        "coloured in float32 and widened (opt-in shortcut, NOT the reference's arithmetic)")
                                     + "; host-coefficient (parity) mode is float64 throughout",
                       "rng_precision": args.rng_precision,
                       "iters_per_step_per_gpu": iters_worker, "kernel_path": {0: "direct", 1: "wave-fft", 2: "chirp-z", 3: "lanes50-fft"}[h.kernel_path()],
                       "launch": {"single": "one process, one GPU", "threads": f"one process, {workers} worker threads",
                                  "ranks": f"{workers} processes (launcher), fast_amd.rendezvous"}[mode],
                       "workers": workers, "devices": devices if mode != "ranks" else "LOCAL_RANK per process",
                       "parallelism": f"realisations sharded over {workers} worker(s) on {n_devices} GPU(s)", "result_exchange": exchange_name(),
                       "rccl_ranks": rccl_ranks,
                       "rccl_required": (rccl_problem or "met") if args.require_rccl else None,
                       "histogram_total": None if hist_total is None else int(np.sum(hist_total))},
            "roofline": roofline(args, N, Np, tim if (host_cost is None or args.no_pipeline) else host_cost["tim"], args.steps, workers, iters_worker, kernels),
            "launch_plan": launch_plan,
            "pipeline": {"steps_in_flight": 1 if args.no_pipeline else 2,
                         # what K separate steps cost beyond the device-limited time of the same work: launches, the exchange's host side,
                         # result copies, Python -- hidden behind the device's work when two steps are in flight
                         "host_ms_per_step": None if host_cost is None else host_cost["host_ms_per_step"],
                         "one_call_ms_per_step": None if host_cost is None else host_cost["one_call_ms_per_step"],
                         "host_ms_note": "ms_per_step minus the time per step of the same realisations issued as ONE call per worker (device-limited); "
                                         "rows_ms / cols_ms below are HIP-event sums of the timed steps (with two steps in flight consecutive launches' "
                                         "events overlap by a few per cent); roofline.avg_launch_ms is from the one-call pass",
                         "gpu_busy_ms_per_step_per_worker": gpu_ms / args.steps / workers,
                         "gpu_busy_ms_per_step": {"min_worker": float(busy.mean(0).min()), "max_worker": float(busy.mean(0).max()),
                                                  "per_worker": [float(x) for x in busy.mean(0)]},
                         # per step and worker, from the ONE-CALL pass when there is one (nothing else in the queue: the sum of
                         # its launches' HIP-event times IS the kernels' time; profiles/*summary.md reproduces it launch by launch);
                         # the timed steps' own event sums, which overlap with two steps in flight, are kept beside them
                         "rows_ms": clean["rows_ms"] / args.steps / workers,
                         "cols_ms": clean["cols_ms"] / args.steps / workers, "finalize_ms": clean["finalize_ms"] / args.steps / workers,
                         "kernel_ms_from": "one-call pass" if (host_cost is not None and not args.no_pipeline) else "timed steps",
                         "rows_ms_timed_steps_event_sum": tim["rows_ms"] / args.steps / workers,
                         "cols_ms_timed_steps_event_sum": tim["cols_ms"] / args.steps / workers,
                         "init_s": init_s, "powerspec_kernel_ms_warm": sim.powerspec_kernel_ms},
        }
        if clocks:
            ghz = float(np.mean([c[0] for c in clocks]))
            line["clock"] = {"effective_GHz": ghz, "per_worker_GHz": [c[0] for c in clocks], "stamp_span_us": float(np.mean([c[1] for c in clocks])),
                             "nominal_GHz": NOMINAL_GHZ,
                             "how": "fastmc_last_clock: s_memtime against s_memrealtime inside one workgroup in the middle of the last row-kernel "
                                    "launch of the timed steps (every CU busy with the same kernel)"}
            rl = line["roofline"]
            if rl.get("frac") is not None:
                rl["frac_at_effective_clock"] = rl["frac"] * NOMINAL_GHZ / ghz
            if rl.get("issue"):
                rl["issue"]["measured_cycles_per_row_at_effective_clock"] = rl["issue"]["measured_cycles_per_row_at_2.4GHz"] * ghz / NOMINAL_GHZ
        if other:
            key, obj = ("value_f32_draw", "f32_draw") if other_prec == "f32" else ("value_f64_generator", "f64_generator")
            line[key] = total_iters / other["dt"]
            line[obj] = {
                "dtype": "f64 (f32 draw)" if other_prec == "f32" else "f64", "ms_per_step": other["dt"] / args.steps * 1e3,
                "ratio_to_value": (total_iters / other["dt"]) / value,
                "what": ("the same steps with GPU_RNG_PRECISION 'f32' (opt-in): float32 normals (24-bit uniforms, hardware log / sqrt / sin / cos), "
                         "float32 colouring, widened -- NOT the reference's arithmetic; complex128 transform as the headline" if other_prec == "f32" else
                         "the same steps with GPU_RNG_PRECISION 'f64': 53-bit normals, float64 log / sqrt / sincos (fast_amd/csrc/fmc_gen64.h), "
                         "float64 colouring, fused into the row kernel -- the reference's arithmetic end to end"),
                "rows_ms": other["tim"]["rows_ms"] / args.steps / workers, "cols_ms": other["tim"]["cols_ms"] / args.steps / workers,
                "roofline": roofline(args, N, Np, other["tim"] if args.no_pipeline else other["tim_one"], args.steps if args.no_pipeline else 1,
                                     workers, iters_worker, other["kernels"])}
        if exchange is not None:
            line["exchange"] = exchange
        if sustained:
            line["sustained"] = sustained
        if mode == "single" and not args.no_extras and not strong:
            line["extras"] = extras(args, devices[0])
        if mode == "single" and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sim)
            line["speedup_vs_cpu_1core"] = value / line["cpu_baseline"]["value"]
    sync_all()
    # N > 1: configs[3] over all the GPUs (strong scaling) and the configs[4] sweep dealt over them, in this launch form.  Collective:
    # every rank runs it; rank 0 holds the line back until it is done.
    multi_extras, extras_stuck = None, False
    if workers > 1 and not args.no_extras and not strong:
        # the headline is measured: a side measurement must neither lose it nor hang the run -- it runs under a deadline, and when
        # it does not come back the line goes out without it and the process leaves (the other ranks' deadlines do the same)
        ok, val = dist.call_with_deadline(lambda: extras_multi(args, mode, devices, rank, world, rdzv),
                                          float(os.environ.get("FASTMC_BENCH_EXTRAS_TIMEOUT", "300")))
        multi_extras = val if ok else {"error": str(val)}
        extras_stuck = (not ok) and bool(dist.stuck_threads())
    if rank == 0:
        if multi_extras is not None:
            line["extras_multi_gpu"] = multi_extras
        try:      # RCCL prints a version banner through C stdio: flush it first so that the JSON line is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if not extras_stuck:            # (a rank whose side measurement is stuck inside a collective cannot meet the others again)
        sync_all()
    if rccl_problem:
        # NOT a clean exit: with a thread still inside RCCL the exit hook of fast_amd.dist leaves through os._exit with the status
        # recorded here (a raised SystemExit passes none of its hooks and used to come out as 0: ADVICE r5)
        if rank == 0:
            print(rccl_problem, file=sys.stderr, flush=True)
        dist.mark_exit(3)
        sys.exit(3)
    # the line is out: if a thread is still blocked inside RCCL although its communicator was aborted, fast_amd.dist's exit hook
    # skips the runtime teardown -- with status 0 only because the run got here (an exception on the way exits non-zero)
    dist.mark_clean_exit()


if __name__ == "__main__":
    main()
