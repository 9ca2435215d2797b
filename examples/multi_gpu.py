#!/usr/bin/env python3
"""One run sharded over the GPUs of a node.  Two ways, neither needs torch:

    python examples/multi_gpu.py --gpus 8                  one process, 8 handles on 8 threads (GPU_DEVICES)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        examples/multi_gpu.py                              one process per GPU (any launcher that sets RANK /
                                                           WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)

Either way every process gets the full result vector, identical to a single-GPU run with the same SEED: the device
generator is keyed on the global iteration index, and the per-GPU results are exchanged once (RCCL inside
libfastmc.so; through the host when RCCL cannot initialise).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np               # noqa: E402
import fast                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
args = ap.parse_args()
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))

h, cn2, wind = fast.turbulence_models.HV57_Bufton_profile(4)
params = {
    "NPXLS": 2048, "DX": 0.01, "NITER": 100000 // (2 * max(world, args.gpus)) * 2 * max(world, args.gpus), "NCHUNKS": 10, "SEED": 3,
    "LOGLEVEL": "ERROR", "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": wind, "FFTW": True,
    "WIND_DIR": np.array([0., 90., 180., 270.]), "ZENITH_ANGLE": 55, "AO_MODE": "AO", "DSUBAP": 0.1,
}
if world == 1:
    params["GPU_DEVICES"] = list(range(args.gpus))     # this process drives them all
# else: GPU_DEVICE defaults to LOCAL_RANK and GPU_SHARD 'auto' shards over the ranks of the launch
sim = fast.Fast(params)
res = sim.run()
if rank == 0:
    how = sim._group.exchange if world == 1 else sim._tr.name
    print(res, f"exchange: {how}; dB histogram total: {int(sim.histogram().sum())}")
