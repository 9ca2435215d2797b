#!/usr/bin/env python3
"""Per-step host cost of the sharded step, measured on ONE device:

  A  1 worker x 10 000 iterations/step          the headline job
  B  8 workers x 1 250 iterations/step          the same device work cut into eight handles on eight streams: what eight workers cost
  C  1 worker x 1 250 iterations/step           one worker's share alone: wall - kernels = the fixed cost of a step

each with one step at a time (--no-pipeline: enqueue, exchange, wait -- rounds 1-3) and with two steps in flight (default).
    python tools/host_overhead.py [steps] > profiles/r04_host_overhead_workers.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def bench(workers, iters, pipeline):
    env = dict(os.environ)
    env["FASTMC_BENCH_DEVICES"] = ",".join(["0"] * workers)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(workers), "--steps", str(steps), "--warmup", "3", "--iters-per-step", str(iters),
           "--no-cpu-baseline", "--no-extras", "--no-sustained", "--no-f64-generator-pass"] + ([] if pipeline else ["--no-pipeline"])
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    if r.returncode:
        raise SystemExit(r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


print(f"# host cost per step on one MI355X, {steps} timed steps each (tools/host_overhead.py)")
print(f"{'case':46s} {'steps in flight':>15s} {'wall ms/step':>13s} {'same work as ONE call, ms/step':>35s} {'wall - one call':>15s} {'per worker':>11s} {'it/s':>10s}")
rows = {}
for tag, w, it in (("A 1 worker x 10000 it", 1, 10000), ("B 8 workers x 1250 it (one device)", 8, 1250), ("C 1 worker x 1250 it", 1, 1250)):
    for pipe in (False, True):
        d = bench(w, it, pipe)
        wall = d["ms_per_step"]
        ker = d["pipeline"]["one_call_ms_per_step"]          # the same realisations as ONE call per worker: device-limited
        rows[(tag, pipe)] = (wall, ker)
        print(f"{tag:46s} {2 if pipe else 1:15d} {wall:13.3f} {ker:35.3f} {wall - ker:15.3f} {(wall - ker) / w:11.3f} {d['value']:10.0f}")
a1, a2 = rows[("A 1 worker x 10000 it", False)][0], rows[("A 1 worker x 10000 it", True)][0]
b1, b2 = rows[("B 8 workers x 1250 it (one device)", False)][0], rows[("B 8 workers x 1250 it (one device)", True)][0]
print(f"\\n# eight workers against one on the same device work: one step at a time {b1 - a1:+.3f} ms/step = {(b1 - a1) / 8:+.3f} per worker; "
      f"two in flight {b2 - a2:+.3f} ms/step = {(b2 - a2) / 8:+.3f} per worker")
print("# ('one call' = the timed steps' realisations issued as a single fastmc_run per worker: the device-limited time of the same work; "
      "wall - one call is what issuing, exchanging and collecting the steps one by one costs)")
