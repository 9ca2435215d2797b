"""Sharding of the Monte-Carlo iterations over GPUs and the one exchange at the end of a run.

Iterations are independent given the spectrum (fast/fast.py:130-134 loops over chunks that 589-605 draws afresh,
647-668 reduces each screen on its own), and the device generator is keyed on the GLOBAL realisation index, so a
range [real0, real0 + n) can be cut anywhere: shard r computes its contiguous piece and the concatenation is
identical, bit for bit, to a single-GPU run.  Two ways to drive several GPUs, both torch-free:

  one process, N devices      fast_amd/multi.py: DeviceGroup (N handles on N threads; `GPU_DEVICES: [0, 1, ...]`)
  one process per GPU         a launcher sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT
                              (`python -m torch.distributed.run`, srun, mpirun ...); fast_amd/rendezvous.py connects
                              the ranks; `Fast.run()` shards when GPU_SHARD allows it.

Transports of the final exchange (same interface: .world, .rank, .gather(local_values, handle), .reduce_hist(hist)):
  RcclTransport  -- RCCL inside libfastmc.so: ncclAllGather of the powers and ncclAllReduce of the histogram on the
                    device buffers over xGMI (fastmc_comm_gather); the 128-byte unique id travels through the rendezvous.
  HostTransport  -- the same exchange through the rendezvous sockets on the host copies fastmc_run already returned
                    (8 B per iteration): the fall-back when RCCL cannot initialise, and what the CPU tests run.
`make_transport` decides between them COLLECTIVELY: every rank reports whether its communicator came up and all take
the host path unless all succeeded, so no rank is ever left waiting in a collective the others skipped.
"""
import logging
import os
import threading

import numpy as np

logger = logging.getLogger(__name__)


def shard_ranges(n_real_total, world):
    """[(real0, n)] per shard: contiguous, sizes differing by at most one."""
    base, extra = divmod(int(n_real_total), int(world))
    out, r0 = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((r0, n))
        r0 += n
    return out


def shard_range(n_real_total, world, rank):
    """This rank's (real0, n): contiguous, equal ranges (the RCCL all-gather needs equal counts, so world must divide)."""
    if n_real_total % world != 0:
        raise Exception(f"number of realisations ({n_real_total}) must be a multiple of the number of GPUs ({world})")
    n = n_real_total // world
    return rank * n, n


def assemble(parts, complex_out=False):
    """[shard][Re block | Im block] -> [Re of all realisations | Im of all realisations] (the order fastmc_run uses
    for a single range).  `parts`: one 1-D array per shard (float64 powers, or complex128 amplitudes / their float64
    pairs when complex_out), possibly of different lengths."""
    re, im = [], []
    for p in parts:
        p = np.ascontiguousarray(p)
        if complex_out and not np.iscomplexobj(p):
            p = p.view(np.complex128)
        n = p.size // 2
        re.append(p[:n])
        im.append(p[n:])
    return np.concatenate(re + im)


class HostTransport:
    """Exchange through the rendezvous sockets (host memory)."""
    name = "host"

    def __init__(self, rdzv):
        self.rdzv, self.world, self.rank = rdzv, rdzv.world, rdzv.rank

    def gather(self, local, handle=None):
        local = np.ascontiguousarray(local)
        cplx = np.iscomplexobj(local)
        parts = self.rdzv.exchange(local.tobytes())
        dt = np.complex128 if cplx else np.float64
        return [np.frombuffer(p, dtype=dt) for p in parts]

    def reduce_hist(self, local_hist):
        return self.rdzv.all_reduce(np.asarray(local_hist, dtype=np.int64), "sum")


class RcclTransport:
    """The exchange runs inside libfastmc.so on the handle's device buffers (fastmc_comm_gather)."""
    name = "rccl"

    def __init__(self, rdzv):
        self.rdzv, self.world, self.rank = rdzv, rdzv.world, rdzv.rank

    def gather(self, local, handle):
        local = np.ascontiguousarray(local)
        cplx = np.iscomplexobj(local)
        nval = local.size * (2 if cplx else 1)                   # float64 values resident on the device
        allp, _ = handle.comm_gather(nval, self.world, None)
        allp = allp.reshape(self.world, nval)
        return [allp[r].view(np.complex128) if cplx else allp[r] for r in range(self.world)]

    def gather_with_hist(self, local, handle, hist_range):
        """Powers and the dB histogram of the handle's last run in one call (ncclAllGather + ncclAllReduce)."""
        nval = np.asarray(local).size * (2 if np.iscomplexobj(local) else 1)
        allp, hist = handle.comm_gather(nval, self.world, hist_range)
        return allp.reshape(self.world, nval), hist

    def device_hist(self, handle, hist_range):
        """Global dB histogram of the handle's last run: histogram kernel + ncclAllReduce(sum, uint64) on the device."""
        return handle.comm_gather(1, self.world, hist_range, powers=False)[1]

    def reduce_hist(self, local_hist):
        return self.rdzv.all_reduce(np.asarray(local_hist, dtype=np.int64), "sum")


_TRANSPORT = {}          # device -> transport of this process (the communicator belongs to the device)


def make_transport(handle, rdzv, rccl_timeout=None):
    """The transport for this process's device: RCCL when EVERY rank's communicator initialises, else the host path.
    The decision is collective (two rendezvous exchanges); FASTMC_DISABLE_RCCL=1 forces the host path."""
    from . import _lib
    key = handle.device
    if key in _TRANSPORT:
        return _TRANSPORT[key]
    timeout = float(os.environ.get("FASTMC_RCCL_TIMEOUT", "90")) if rccl_timeout is None else rccl_timeout
    # 1) rank 0 creates the unique id; everybody learns whether that worked
    msg = b"\x00"
    if rdzv.rank == 0:
        try:
            msg = b"\x01" + _lib.comm_unique_id()
        except Exception as e:                      # library missing, librccl missing, disabled ...
            msg = b"\x00" + str(e).encode()
    msg = rdzv.broadcast(msg, src=0)
    why = ""
    ok = msg[:1] == b"\x01"
    if not ok:
        why = msg[1:].decode(errors="replace")
    else:
        # 2) every rank initialises its communicator under a timeout and reports; all-or-nothing
        box = {}

        def _init():
            try:
                handle.comm_init(msg[1:], rdzv.world, rdzv.rank)
                box["ok"] = True
            except Exception as e:
                box["err"] = str(e)
        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(timeout)
        mine = bool(box.get("ok"))
        flags = rdzv.all_gather_array(np.array([1 if mine else 0], dtype=np.int32)).ravel()
        ok = bool(flags.all())
        if not ok:
            why = box.get("err") or ("ncclCommInitRank did not return in time" if not mine else
                                     f"rank(s) {np.flatnonzero(flags == 0).tolist()} failed")
            if mine:       # peers may be gone: never wait for the teardown of a half-built clique
                threading.Thread(target=lambda: handle.comm_destroy(), daemon=True).start()
    if ok:
        tr = RcclTransport(rdzv)
    else:
        if rdzv.rank == 0:
            logger.warning(f"RCCL exchange unavailable ({why}); results are exchanged through the host")
        tr = HostTransport(rdzv)
        tr.why = why
    _TRANSPORT[key] = tr
    return tr


def run_sharded(n_real_total, compute_local, transport, handle=None):
    """compute_local(real0, n_local) -> [2*n_local] values ([Re-screen results | Im-screen results]), float64 powers
    or complex128 amplitudes (COHERENT); every rank returns the full [2*n_real_total] vector."""
    real0, n_local = shard_range(n_real_total, transport.world, transport.rank)
    local = np.asarray(compute_local(real0, n_local))
    parts = transport.gather(local, handle)
    return assemble(parts, complex_out=np.iscomplexobj(local))


def histogram_sharded(local_hist, transport):
    return transport.reduce_hist(local_hist)
