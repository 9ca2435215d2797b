"""Multi-GPU sharding of the Monte-Carlo iterations: one process per GPU, launched by
`torch.distributed.run`; results exchanged once per run.

Iterations are independent given the spectrum (fast/fast.py:589-605, 647-668 of the reference),
and the device generator is keyed on the GLOBAL realisation index, so rank r simply computes
realisations [r*n, (r+1)*n) and the concatenation is identical to a single-GPU run.

Transports for the final exchange:
  RcclTransport  -- RCCL inside libfastmc.so (ncclAllGather of the powers, ncclAllReduce of the
                    histogram, on the device buffers over xGMI); the 128-byte unique id is
                    broadcast through the launcher's torch.distributed store.
  TorchTransport -- torch.distributed collectives on host tensors (backend gloo on CPU in the
                    tests; also the stand-in if RCCL cannot initialise).
torch is imported only here and only when a process group exists; the compute path never sees it.

Load order: PyTorch's ROCm wheel bundles its own libamdhip64 / libhsa-runtime64, and the first HIP
runtime loaded serves the whole process.  A program that uses both must `import torch` BEFORE the
first fast_amd call that loads libfastmc.so (torch cannot run on /opt/rocm's newer runtime; the
library runs on either).  `torch.distributed.run` launchers that initialise the process group
first, as bench.py does, satisfy this automatically.
"""
import numpy as np


def shard_range(n_real_total, world, rank):
    """Contiguous, equal ranges; the all-gather needs equal counts, so world must divide."""
    if n_real_total % world != 0:
        raise Exception(f"number of realisations ({n_real_total}) must be a multiple of the number of GPUs ({world})")
    n = n_real_total // world
    return rank * n, n


def assemble(gathered, world, n_local, complex_out=False):
    """[rank][Re block | Im block] -> [Re of all realisations | Im of all realisations]
    (the order fastmc_run uses for a single range).  complex_out: COHERENT runs, every value a
    complex128 amplitude carried as two float64."""
    g = np.asarray(gathered)
    if complex_out:
        g = np.ascontiguousarray(g, dtype=np.float64).reshape(world, -1).view(np.complex128)
    g = g.reshape(world, 2, n_local)
    return np.concatenate([g[:, 0].ravel(), g[:, 1].ravel()])


class TorchTransport:
    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.cuda = dist.get_backend(group) == "nccl"

    def _t(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        return t.cuda() if self.cuda else t

    def all_gather(self, local):
        src = self._t(local)
        bufs = [self.torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(bufs, src, group=self.group)
        return np.stack([b.cpu().numpy() for b in bufs])

    def all_reduce_sum(self, local):
        t = self._t(local)
        self.dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()


class RcclTransport:
    """The exchange runs inside libfastmc.so on the handle's device buffers."""

    def __init__(self, handle, group=None):
        import torch.distributed as dist
        from . import _lib
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        ids = [_lib.comm_unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(ids, src=0, group=group)
        handle.comm_init(ids[0], self.world, self.rank)
        self.handle = handle

    def gather(self, n_local_values, hist_range=None):
        allp, hist = self.handle.comm_gather(n_local_values, self.world, hist_range)
        return allp.reshape(self.world, n_local_values), hist


def run_sharded(n_real_total, compute_local, transport):
    """compute_local(real0, n_local) -> [2*n_local] values ([Re-screen results | Im-screen results]),
    float64 powers or complex128 amplitudes (COHERENT); every rank returns the full [2*n_real_total] vector."""
    real0, n_local = shard_range(n_real_total, transport.world, transport.rank)
    local = np.asarray(compute_local(real0, n_local))
    cplx = np.iscomplexobj(local)
    if isinstance(transport, RcclTransport):
        gathered, _ = transport.gather((4 if cplx else 2) * n_local)      # float64 values on the device
    else:
        gathered = transport.all_gather(np.ascontiguousarray(local).view(np.float64) if cplx else local)
    return assemble(gathered, transport.world, n_local, complex_out=cplx)


def histogram_sharded(local_hist, transport):
    return transport.all_reduce_sum(np.asarray(local_hist, dtype=np.int64))
