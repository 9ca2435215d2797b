#!/usr/bin/env python3
"""BASELINE config 5 on the GPU(s): zenith-angle scan, 32 geometries x 4096 iterations, N = 1024,
AO-corrected residual spectrum.  Single process or one process per GPU under any launcher that sets RANK /
WORLD_SIZE / LOCAL_RANK / MASTER_* (samples are dealt round-robin to the ranks; no torch).  Prints one JSON line
with the wall time split."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_amd  # noqa: E402
from fast_amd import sweep  # noqa: E402


def main():
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
    p = {"NPXLS": 1024, "DX": 0.01, "SEED": 1, "LOGLEVEL": "ERROR", "D_GROUND": 0.8, "H_SAT": 36e6, "H_TURB": h,
         "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": np.array([0., 90., 180., 270.]), "AO_MODE": "AO", "DSUBAP": 0.1,
         "ALIAS": True, "FFTW": True}          # GPU_DEVICE defaults to LOCAL_RANK
    angles = np.linspace(0, 70, 32)
    t0 = time.perf_counter()
    recs = sweep.gather_records(sweep.zenith_scan(p, angles, niter=4096, rank=rank, world=world))
    wall = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"workload": "configs[4]: 32 zenith angles x 4096 iterations, 1024^2, AO+alias", "n_gpus": world,
                          "wall_s": wall, "iterations_per_s": 32 * 4096 / wall,
                          "sum_init_s": sum(r["init_s"] for r in recs), "sum_run_s": sum(r["run_s"] for r in recs),
                          "sum_powerspec_kernel_ms": sum(r["powerspec_kernel_ms"] for r in recs),
                          "mean_dB_rel": [round(r["mean_dB_rel"], 3) for r in recs]}))


if __name__ == "__main__":
    main()
