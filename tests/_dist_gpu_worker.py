"""Worker of test_gpu_dist.py: one of two plain processes (RANK / WORLD_SIZE / MASTER_* in the environment, no torch)
sharing GPU 0, running ONE sharded `Fast.run()`; every rank must get exactly the vector an unsharded run produces.
RCCL refuses two ranks on one device, so the transport decision must fall back to the host exchange on BOTH ranks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fast_amd  # noqa: E402
from fast_amd import rendezvous  # noqa: E402


def main():
    rdzv = rendezvous.from_env()
    rank, world = rdzv.rank, rdzv.world
    h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
    p = {"NPXLS": 512, "DX": 0.01, "NITER": 400, "NCHUNKS": 4, "SEED": None if os.environ.get("NOSEED") else 21,
         "LOGLEVEL": "ERROR", "D_GROUND": 0.4, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w,
         "WIND_DIR": np.array([0., 90., 180., 270.]), "DSUBAP": 0.1, "GPU_DEVICE": 0,
         "COHERENT": bool(os.environ.get("COHERENT"))}
    sim = fast_amd.Fast(dict(p))
    r = sim.run()._r                                   # sharded: GPU_SHARD 'auto' sees the multi-rank environment
    assert r.shape == (400,) and np.isfinite(r).all() and np.iscomplexobj(r) == p["COHERENT"]
    assert sim._tr.name == "host", sim._tr.name        # two ranks on one device: RCCL cannot form the clique
    hist = sim.histogram(-40.0, 10.0, 50)              # of the assembled vector, on the device
    pw = np.abs(r) ** 2 if p["COHERENT"] else r
    assert hist.sum() == 400 and np.array_equal(hist[:50], np.histogram(10 * np.log10(pw), bins=50, range=(-40, 10))[0])
    p1 = dict(p)
    p1.update({"GPU_SHARD": False, "SEED": sim._device_seed})
    single = fast_amd.Fast(p1).run()._r
    assert np.array_equal(r, single), np.abs(r - single).max()
    gathered = rdzv.exchange(r.tobytes())
    assert all(g == gathered[0] for g in gathered)
    assert "torch" not in sys.modules
    rdzv.barrier()
    if rank == 0:
        print("GPU DIST OK", world, sim._tr.name)


if __name__ == "__main__":
    main()
