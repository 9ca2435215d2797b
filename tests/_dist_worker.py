"""Worker of tests/test_dist_gloo.py: run under torch.distributed.run with backend gloo (CPU), the exchange through
examples/torch_transport.py (fast_amd itself is torch-free; tests/test_dist_rdzv.py covers its own rendezvous).
compute_local is the ORACLE fed with the restated device generator, i.e. exactly what each GPU
rank computes; the sharded result must equal the single-process one bit for bit."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
from fast_amd import dist as fd   # noqa: E402
from torch_transport import TorchTransport   # noqa: E402
from oracle import fastref as R, devrng   # noqa: E402

N, Np, SEED, NREAL = 16, 6, 77, 8


def problem():
    g = R.main_grid(N, 0.02)
    ps = R.von_karman(g.fabs, np.array([3e-13]), 25.0, 1e-3).sum(0) * 2 * np.pi * (2 * np.pi / 1550e-9) ** 2 * 1e-3
    W = np.ones((Np, Np))
    return ps, g.df, W


def compute(real0, n, coherent=False):
    ps, df, W = problem()
    coeffs = np.stack([devrng.device_coefficients(SEED, real0 + j, N) for j in range(n)])
    chi = devrng.device_logamp_normals(SEED, 2 * real0, 2 * n) * 0.1
    la = np.concatenate([chi[0::2], chi[1::2]])
    return R.powers_from_coefficients(coeffs, ps, df, W, 0.02, la, coherent=coherent)


def main():
    dist.init_process_group("gloo")
    tr = TorchTransport()
    full = fd.run_sharded(NREAL, compute, tr)
    single = compute(0, NREAL)
    assert np.array_equal(full, single), (full, single)
    full_c = fd.run_sharded(NREAL, lambda r0, n: compute(r0, n, coherent=True), tr)      # COHERENT: complex amplitudes
    single_c = compute(0, NREAL, coherent=True)
    assert np.iscomplexobj(full_c) and np.array_equal(full_c, single_c), (full_c, single_c)
    h_local = np.histogram(10 * np.log10(compute(*fd.shard_range(NREAL, tr.world, tr.rank))), bins=8, range=(-40, 10))[0]
    h_all = fd.histogram_sharded(h_local, tr)
    assert np.array_equal(h_all, np.histogram(10 * np.log10(single), bins=8, range=(-40, 10))[0])
    try:
        fd.shard_range(7, 2, 0)
        raise SystemExit("expected an exception for an indivisible range")
    except Exception as e:
        assert "multiple" in str(e)
    dist.barrier()
    if tr.rank == 0:
        print("DIST OK", tr.world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
