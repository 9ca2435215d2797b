"""Where Fast.__init__ spends its time (second and later objects: caches warm): tools/init_profile.py"""
import argparse, cProfile, pstats, time
import numpy as np
import bench, fast_amd

a = argparse.Namespace(precision="f64", npxls=1024, ao_mode="AO", batch=0)
p = bench.workload_params(a)
p["GPU_DEVICE"] = 0
p["NITER"] = 4096; p["NCHUNKS"] = 1
fast_amd.Fast(dict(p)).run()
ts = []
for z in (10, 20, 30, 40):
    q = dict(p); q["ZENITH_ANGLE"] = z
    t0 = time.perf_counter(); s = fast_amd.Fast(q); t1 = time.perf_counter(); s.run(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
print("init/run seconds:", [(round(a, 4), round(b, 4)) for a, b in ts])
q = dict(p); q["ZENITH_ANGLE"] = 50
pr = cProfile.Profile(); pr.enable(); s = fast_amd.Fast(q); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
