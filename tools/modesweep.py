"""iterations/s at 1024^2 for the option combinations of the Monte-Carlo path: tools/modesweep.py"""
import argparse, time
import numpy as np
import bench, fast_amd

cases = {"base NOAO": {}, "SUBHARM L0=25": {"SUBHARM": True, "L0": 25.0}, "COHERENT": {"COHERENT": True},
         "AO+alias": {"AO_MODE": "AO"}, "AO+SUBHARM": {"AO_MODE": "AO", "SUBHARM": True, "L0": 25.0},
         "f32 SUBHARM": {"SUBHARM": True, "L0": 25.0, "GPU_PRECISION": "f32"}}
for name, over in cases.items():
    a = argparse.Namespace(precision="f64", npxls=1024, ao_mode="NOAO", batch=0)
    p = bench.workload_params(a)
    p.update(over)
    p["GPU_DEVICE"] = 0
    sim = fast_amd.Fast(p)
    sim.run()
    t0 = time.perf_counter()
    for i in range(3):
        sim.run()
    dt = (time.perf_counter() - t0) / 3
    t = sim.timing
    print(f"{name:16s} {p['NITER'] / dt:9.0f} it/s (Fast.run wall)  rows {t['rows_ms']:.2f} cols {t['cols_ms']:.2f} finalize {t['finalize_ms']:.3f} ms")
