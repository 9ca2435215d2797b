#!/usr/bin/env python3
"""One run sharded over the GPUs of a node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 examples/multi_gpu.py

`import torch` comes first (see fast_amd/dist.py: HIP runtime load order).  Every rank gets the full
result vector, identical to a single-GPU run with the same SEED: the device generator is keyed on the
global iteration index, and the per-GPU results are exchanged once by RCCL inside libfastmc.so.
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np               # noqa: E402
import fast                      # noqa: E402

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

h, cn2, wind = fast.turbulence_models.HV57_Bufton_profile(4)
params = {
    "NPXLS": 2048, "DX": 0.01, "NITER": 100000 // dist.get_world_size() * dist.get_world_size(), "NCHUNKS": 10, "SEED": 3,
    "LOGLEVEL": "ERROR", "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": wind,
    "WIND_DIR": np.array([0., 90., 180., 270.]), "ZENITH_ANGLE": 55, "AO_MODE": "AO", "DSUBAP": 0.1,
    "GPU_DEVICE": local_rank,          # GPU_SHARD 'auto' (default) shards because a process group exists
}
sim = fast.Fast(params)
res = sim.run()
if dist.get_rank() == 0:
    print(res, f"exchange: {type(getattr(sim, '_tr', None)).__name__}")
dist.barrier()
dist.destroy_process_group()
