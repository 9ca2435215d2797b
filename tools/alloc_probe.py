"""Cost of the per-object device allocations in sweeps: objects overlapping in lifetime vs freed first."""
import argparse, time, gc
import numpy as np
import bench, fast_amd

a = argparse.Namespace(precision="f64", npxls=1024, ao_mode="AO", batch=0)
p = bench.workload_params(a); p["GPU_DEVICE"] = 0; p["NITER"] = 4096; p["NCHUNKS"] = 1
fast_amd.Fast(dict(p)).run()
for mode in ("overlap", "free-first", "overlap", "free-first"):
    s = None
    t = []
    for z in range(8):
        q = dict(p); q["ZENITH_ANGLE"] = 5 * z
        if mode == "free-first" and s is not None:
            s._handle.close(); s = None; gc.collect()
        t0 = time.perf_counter(); s = fast_amd.Fast(q); t1 = time.perf_counter(); s.run(); t2 = time.perf_counter()
        t.append((t1 - t0, t2 - t1))
    print(mode, "init ms", [round(1e3 * x) for x, _ in t], "run ms", [round(1e3 * y) for _, y in t])
