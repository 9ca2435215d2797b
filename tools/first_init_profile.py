"""Profile of the FIRST Fast() of a process (cold caches, HIP context creation): tools/first_init_profile.py [NPXLS]"""
import argparse, cProfile, pstats, sys, time
t00 = time.perf_counter()
import numpy as np
import bench, fast_amd
t_imp = time.perf_counter() - t00
a = argparse.Namespace(precision="f64", npxls=int(sys.argv[1]) if len(sys.argv) > 1 else 1024, ao_mode="AO", batch=0)
p = bench.workload_params(a)
p["GPU_DEVICE"] = 0
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
sim = fast_amd.Fast(p)
t1 = time.perf_counter(); pr.disable()
r = sim.run(); t2 = time.perf_counter()
print(f"import {t_imp:.2f} s, first Fast() {t1 - t0:.3f} s, first run() of {p['NITER']} iterations {t2 - t1:.3f} s")
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
