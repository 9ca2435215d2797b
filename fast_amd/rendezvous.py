"""Rendezvous of the ranks of a one-process-per-GPU run: plain sockets, no torch, no MPI.

A launcher (`python -m torch.distributed.run`, `srun`, `mpirun`, a shell loop) starts one process per GPU and
tells each its place through the usual environment: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT.  This
module turns that into the three small host-side collectives the Monte-Carlo path needs once per run --
broadcast (the 128-byte RCCL unique id, an entropy seed), all-gather (success flags, timing, the host fall-back of the
result exchange) and barrier -- over one star of stream sockets centred on rank 0.  Message sizes are bytes to a few
hundred kilobytes (8 B per iteration), so latency, not bandwidth, is what it costs: ~0.1 ms per collective on one node.

The data path itself never comes here when RCCL is available: the per-iteration powers and the dB histogram are
exchanged on the device buffers by `fastmc_comm_gather` (fast_amd/dist.py: RcclTransport).

Endpoint, in this order:
  FASTMC_RDZV = "tcp://host:port" | "unix:name"            explicit
  one node (LOCAL_WORLD_SIZE == WORLD_SIZE, or MASTER_ADDR is a loopback address):
        abstract unix socket "fastmc-rdzv-<MASTER_PORT>-<run id>"  (no TCP port to collide with the launcher's own store)
  otherwise: tcp://MASTER_ADDR:(MASTER_PORT + 1)

The reference (ojdf/fast) is single-process; this has no counterpart there.
"""
import atexit
import os
import socket
import struct
import time

import numpy as np


class RendezvousError(RuntimeError):
    pass


def env_world():
    """(rank, world, local_rank) from the launcher's environment; (0, 1, 0) when not launched as one of several ranks."""
    try:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
    except ValueError as e:
        raise RendezvousError(f"bad RANK / WORLD_SIZE / LOCAL_RANK in the environment: {e}")
    if world < 1 or not 0 <= rank < world:
        raise RendezvousError(f"RANK={rank} outside WORLD_SIZE={world}")
    return rank, world, local


def _endpoint(world):
    spec = os.environ.get("FASTMC_RDZV")
    if spec:
        if spec.startswith("unix:"):
            return "unix", spec[5:]
        if spec.startswith("tcp://"):
            host, _, port = spec[6:].rpartition(":")
            return "tcp", (host, int(port))
        raise RendezvousError("FASTMC_RDZV must be tcp://host:port or unix:name")
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29400"))
    one_node = os.environ.get("LOCAL_WORLD_SIZE") == str(world) or addr in ("127.0.0.1", "localhost", "::1")
    if one_node and hasattr(socket, "AF_UNIX"):
        run_id = os.environ.get("TORCHELASTIC_RUN_ID", "") or os.environ.get("SLURM_JOB_ID", "")
        return "unix", f"fastmc-rdzv-{port}-{run_id}"
    return "tcp", (addr, port + 1)


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise RendezvousError("peer closed the rendezvous connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class Rendezvous:
    """Star of sockets centred on rank 0.  Every method is a collective: all ranks call it, in the same order."""

    def __init__(self, rank, world, kind, address, timeout=120.0, io_timeout=None):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        # `timeout` bounds the meeting of the ranks; a collective later on waits for the slowest rank's Monte-Carlo shard,
        # which may take much longer (a dead peer closes its socket and is noticed at once): FASTMC_RDZV_IO_TIMEOUT, 1 h
        self.io_timeout = float(os.environ.get("FASTMC_RDZV_IO_TIMEOUT", "3600")) if io_timeout is None else float(io_timeout)
        self._peers = []          # rank 0: socket of rank r at index r - 1
        self._up = None           # other ranks: socket to rank 0
        self._listener = None
        fam = socket.AF_UNIX if kind == "unix" else socket.AF_INET
        addr = ("\0" + address) if kind == "unix" else address
        self.endpoint = f"{kind}:{address}"
        if self.world == 1:
            return
        if self.rank == 0:
            ls = socket.socket(fam, socket.SOCK_STREAM)
            if kind == "tcp":
                ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                ls.bind(addr)
            except OSError as e:
                raise RendezvousError(f"rank 0 cannot bind the rendezvous endpoint {self.endpoint}: {e}")
            ls.listen(self.world)
            ls.settimeout(self.timeout)
            self._listener = ls
            peers = {}
            try:
                while len(peers) < self.world - 1:
                    c, _ = ls.accept()
                    c.settimeout(self.timeout)
                    if kind == "tcp":
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    (r, w) = struct.unpack("<ii", _recv(c))
                    if w != self.world or not 0 < r < self.world or r in peers:
                        raise RendezvousError(f"unexpected hello (rank {r} of {w}) at {self.endpoint}")
                    peers[r] = c
            except socket.timeout:
                raise RendezvousError(f"only {len(peers) + 1} of {self.world} ranks reached {self.endpoint} in {self.timeout:.0f} s")
            self._peers = [peers[r] for r in range(1, self.world)]
            for c in self._peers:
                c.settimeout(self.io_timeout)
        else:
            deadline = time.monotonic() + self.timeout
            last = None
            while True:
                s = socket.socket(fam, socket.SOCK_STREAM)
                try:
                    s.connect(addr)
                    break
                except OSError as e:      # rank 0 is not listening yet
                    last = e
                    s.close()
                    if time.monotonic() > deadline:
                        raise RendezvousError(f"rank {self.rank} cannot reach rank 0 at {self.endpoint}: {last}")
                    time.sleep(0.05)
            s.settimeout(self.timeout)
            if kind == "tcp":
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(s, struct.pack("<ii", self.rank, self.world))
            s.settimeout(self.io_timeout)
            self._up = s

    # ---- the one primitive: all-gather of byte strings
    def exchange(self, payload):
        """Every rank contributes `payload` (bytes); every rank gets the list of all ranks' payloads, by rank."""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        try:
            if self.rank == 0:
                parts = [payload] + [_recv(c) for c in self._peers]
                blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
                for c in self._peers:
                    _send(c, blob)
                return parts
            _send(self._up, payload)
            blob = _recv(self._up)
        except (OSError, socket.timeout) as e:
            raise RendezvousError(f"rendezvous exchange failed on rank {self.rank}: {e}")
        parts, o = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, o)
            parts.append(blob[o + 8:o + 8 + n])
            o += 8 + n
        return parts

    def barrier(self):
        self.exchange(b"")

    def broadcast(self, payload, src=0):
        """`payload` of rank `src` on every rank (other ranks may pass None)."""
        return self.exchange(bytes(payload) if self.rank == src else b"")[src]

    def all_gather_array(self, a):
        """(world, ...) array of every rank's equally shaped array."""
        a = np.ascontiguousarray(a)
        parts = self.exchange(a.tobytes())
        if any(len(p) != a.nbytes for p in parts):
            raise RendezvousError("all_gather_array: ranks contributed arrays of different sizes")
        return np.stack([np.frombuffer(p, dtype=a.dtype).reshape(a.shape) for p in parts])

    def all_reduce(self, a, op="sum"):
        g = self.all_gather_array(a)
        return {"sum": g.sum(0), "max": g.max(0), "min": g.min(0)}[op].astype(np.asarray(a).dtype)

    def close(self):
        for s in self._peers + [self._up, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._up, self._listener = [], None, None


_GLOBAL = None


def from_env(timeout=None):
    """The process-wide rendezvous of a multi-rank launch (created on first use), or None when WORLD_SIZE <= 1."""
    global _GLOBAL
    if _GLOBAL is not None:
        return _GLOBAL
    rank, world, _ = env_world()
    if world <= 1:
        return None
    kind, address = _endpoint(world)
    t = float(os.environ.get("FASTMC_RDZV_TIMEOUT", "120")) if timeout is None else timeout
    _GLOBAL = Rendezvous(rank, world, kind, address, t)
    atexit.register(_GLOBAL.close)
    return _GLOBAL
