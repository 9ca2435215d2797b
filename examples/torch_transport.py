"""Optional: the result exchange of fast_amd/dist.py over a torch.distributed process group.

fast_amd itself never imports torch (its own rendezvous, fast_amd/rendezvous.py, needs only the launcher's
environment).  A program that already runs a torch.distributed group can hand this adapter to
`fast_amd.dist.run_sharded` instead; tests/test_dist_gloo.py runs it with the gloo backend on CPU.
"""
import numpy as np


class TorchTransport:
    name = "torch.distributed"

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def gather(self, local, handle=None):
        local = np.ascontiguousarray(local)
        cplx = np.iscomplexobj(local)
        src = self.torch.from_numpy(local.view(np.float64) if cplx else local)
        bufs = [self.torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(bufs, src, group=self.group)
        out = [b.numpy() for b in bufs]
        return [o.view(np.complex128) for o in out] if cplx else out

    def reduce_hist(self, local_hist):
        t = self.torch.from_numpy(np.ascontiguousarray(local_hist, dtype=np.int64).copy())
        self.dist.all_reduce(t, group=self.group)
        return t.numpy()
