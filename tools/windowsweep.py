"""iterations/s at 1024^2 for several pupil-window sizes (D_GROUND / DX + 2): tools/windowsweep.py [D ...]"""
import argparse, os, sys, time
import numpy as np
import bench, fast_amd

for D in [float(x) for x in sys.argv[1:]] or [0.3, 0.6, 0.8, 1.2, 1.5, 2.0, 2.5]:
    a = argparse.Namespace(precision="f64", npxls=int(os.environ.get("NPX", "1024")), ao_mode="NOAO", batch=0)
    p = bench.workload_params(a)
    p["D_GROUND"] = D
    p["GPU_DEVICE"] = 0
    sim = fast_amd.Fast(p)
    h = sim._handle
    n = 2000
    h.run(1, 0, n, None, float(sim.logamp_var), False)
    t0 = time.perf_counter()
    for i in range(3):
        h.run(1, 0, n, None, float(sim.logamp_var), False)
    dt = (time.perf_counter() - t0) / 3
    t = h.last_timing()
    print(f"N={sim.Npxls} D={D} Np={sim.Npxls_pup} path={'wave' if h.kernel_path() == 1 else 'direct'} {2 * n / dt:.0f} it/s  rows {t['rows_ms']:.2f} cols {t['cols_ms']:.2f} ms")
