"""Worker of tests/test_dist_rdzv.py: one of WORLD_SIZE plain processes (no torch, no launcher beyond the environment).
compute_local is the ORACLE fed with the restated device generator, i.e. exactly what each GPU rank computes; the
sharded result must equal the single-process one bit for bit, through fast_amd's own rendezvous and HostTransport."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_amd import dist as fd, rendezvous   # noqa: E402
from oracle import fastref as R, devrng   # noqa: E402

N, Np, SEED = 16, 6, 77
NREAL = 12


def problem():
    g = R.main_grid(N, 0.02)
    ps = R.von_karman(g.fabs, np.array([3e-13]), 25.0, 1e-3).sum(0) * 2 * np.pi * (2 * np.pi / 1550e-9) ** 2 * 1e-3
    return ps, g.df, np.ones((Np, Np))


def compute(real0, n, coherent=False):
    ps, df, W = problem()
    coeffs = np.stack([devrng.device_coefficients(SEED, real0 + j, N) for j in range(n)])
    chi = devrng.device_logamp_normals(SEED, 2 * real0, 2 * n) * 0.1
    la = np.concatenate([chi[0::2], chi[1::2]])
    return R.powers_from_coefficients(coeffs, ps, df, W, 0.02, la, coherent=coherent)


class FakeHandle:
    """Stands in for _lib.Handle where no GPU exists: the communicator can never come up."""
    device = 0

    def comm_init(self, uid, world, rank):
        raise RuntimeError("no device in this test")

    def comm_destroy(self):
        pass


class StallingHandle:
    """A handle whose device exchange never answers on the ranks in `stall_ranks` (and answers garbage on the others):
    what a hung RCCL collective looks like to fast_amd/dist.py.  run_async / wait / histogram are the oracle."""
    device = 1

    def __init__(self, rank, stall_ranks):
        import threading
        self.rank, self.stall_ranks, self.aborted, self._ev = rank, stall_ranks, 0, threading.Event()

    def run_async(self, seed, real0, n, logamp_var, coherent):
        self._local = compute(real0, n, coherent)

    def wait(self):
        return self._local

    def run(self, seed, real0, n, logamp, logamp_var, coherent):
        return compute(real0, n, coherent)

    def comm_gather(self, nval, world, hist_range=None, powers=True):
        if self.rank in self.stall_ranks:
            self._ev.wait()                      # until comm_abort
            raise RuntimeError("aborted")
        return np.full(nval * world, -1.0), (None if hist_range is None else np.zeros(hist_range[2] + 2, dtype=np.int64))

    def comm_abort(self):
        self.aborted += 1
        self._ev.set()

    def last_exchange_ms(self):
        return 0.0

    def histogram(self, lo, hi, nbins):
        pw = self._local if not np.iscomplexobj(self._local) else np.abs(self._local) ** 2
        h = np.zeros(nbins + 2, dtype=np.int64)
        h[:nbins] = np.histogram(10 * np.log10(pw), bins=nbins, range=(lo, hi))[0]
        return h


def main():
    assert "torch" not in sys.modules
    rdzv = rendezvous.from_env()
    rank, world = rdzv.rank, rdzv.world
    # primitives
    got = rdzv.exchange(bytes([rank]) * (rank + 1))
    assert got == [bytes([r]) * (r + 1) for r in range(world)]
    assert rdzv.broadcast(b"seed-%d" % rank, src=world - 1) == b"seed-%d" % (world - 1)
    assert np.array_equal(rdzv.all_gather_array(np.arange(3) + rank), np.arange(3)[None] + np.arange(world)[:, None])
    assert rdzv.all_reduce(np.array([rank + 1.5]), "max")[0] == world + 0.5
    assert rdzv.all_reduce(np.array([1, rank], dtype=np.int64), "sum").tolist() == [world, world * (world - 1) // 2]
    big = np.random.default_rng(rank).normal(size=200000)                       # > one socket buffer
    assert all(np.array_equal(p, np.random.default_rng(r).normal(size=200000)) for r, p in enumerate(rdzv.all_gather_array(big)))
    rdzv.barrier()
    # the collective transport decision: no rank has a device, so EVERY rank must end on the host path, together
    tr = fd.make_transport(FakeHandle(), rdzv, rccl_timeout=20)
    assert isinstance(tr, fd.HostTransport), type(tr)
    assert fd.make_transport(FakeHandle(), rdzv) is tr                          # cached per device
    # sharded run == single process, bit for bit (powers and COHERENT amplitudes)
    full = fd.run_sharded(NREAL, compute, tr)
    single = compute(0, NREAL)
    assert np.array_equal(full, single), (full, single)
    full_c = fd.run_sharded(NREAL, lambda r0, n: compute(r0, n, coherent=True), tr)
    assert np.iscomplexobj(full_c) and np.array_equal(full_c, compute(0, NREAL, coherent=True))
    h_local = np.histogram(10 * np.log10(compute(*fd.shard_range(NREAL, world, rank))), bins=8, range=(-40, 10))[0]
    assert np.array_equal(fd.histogram_sharded(h_local, tr), np.histogram(10 * np.log10(single), bins=8, range=(-40, 10))[0])
    # a device exchange that hangs on ONE rank: the deadline passes there, the verdict is collective, EVERY rank aborts its
    # communicator and the step finishes on the host sockets with the right vector; later steps stay on the host
    os.environ["FASTMC_EXCHANGE_TIMEOUT"] = "1.5"
    sh = StallingHandle(rank, stall_ranks={world - 1})
    rt = fd.RcclTransport(rdzv, world)
    full2, hist2, info = fd.step_sharded(sh, rt, SEED, 0, NREAL, 0.01, False, (-40.0, 10.0, 8))
    assert np.array_equal(full2, single) and info["exchange"] == "host" and rt.name == "host" and sh.aborted == 1, (info, rt.name)
    assert "given up" in rt.why and rt.rccl_ranks == 0
    assert np.array_equal(hist2[:8], np.histogram(10 * np.log10(single), bins=8, range=(-40, 10))[0])
    full3, _, info3 = fd.step_sharded(sh, rt, SEED, 0, NREAL, 0.01, True)
    assert np.array_equal(full3, compute(0, NREAL, coherent=True)) and info3["exchange"] == "host" and sh.aborted == 1
    # no rank stalls: the device path's answer is taken (here the fake's marker values)
    ok_h, ok_t = StallingHandle(rank, stall_ranks=set()), fd.RcclTransport(rdzv, world)
    full4, _, info4 = fd.step_sharded(ok_h, ok_t, SEED, 0, NREAL, 0.01, False)
    assert info4["exchange"] == "rccl" and ok_t.name == "rccl" and (full4 == -1.0).all() and ok_h.aborted == 0
    assert not fd.stuck_threads()
    try:
        fd.shard_range(7, 2, 0)
        raise SystemExit("expected an exception for an indivisible range")
    except Exception as e:
        assert "multiple" in str(e)
    rdzv.barrier()
    if rank == 0:
        print("RDZV OK", world, rdzv.endpoint.split(":")[0], tr.name)


if __name__ == "__main__":
    main()
