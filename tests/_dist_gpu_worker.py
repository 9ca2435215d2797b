"""Worker of test_gpu_dist.py: two ranks (torch.distributed, gloo) sharing GPU 0 run ONE sharded
`Fast.run()`; every rank must get exactly the vector an unsharded run produces."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fast_amd  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
    p = {"NPXLS": 512, "DX": 0.01, "NITER": 400, "NCHUNKS": 4, "SEED": None if os.environ.get("NOSEED") else 21,
         "LOGLEVEL": "ERROR", "D_GROUND": 0.4, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w,
         "WIND_DIR": np.array([0., 90., 180., 270.]), "DSUBAP": 0.1, "GPU_DEVICE": 0,
         "COHERENT": bool(os.environ.get("COHERENT"))}
    sim = fast_amd.Fast(dict(p))
    r = sim.run()._r                                   # sharded: GPU_SHARD 'auto' sees the process group
    assert r.shape == (400,) and np.isfinite(r).all() and np.iscomplexobj(r) == p["COHERENT"]
    p1 = dict(p)
    p1.update({"GPU_SHARD": False, "SEED": sim._device_seed})
    single = fast_amd.Fast(p1).run()._r
    assert np.array_equal(r, single), np.abs(r - single).max()
    gathered = [None] * world
    dist.all_gather_object(gathered, r.tobytes())
    assert all(g == gathered[0] for g in gathered)
    dist.barrier()
    if rank == 0:
        print("GPU DIST OK", world, type(sim._tr).__name__)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
