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
  FASTMC_RDZV = "tcp://host:port" | "unix:name-or-path"    explicit
  one node (LOCAL_WORLD_SIZE == WORLD_SIZE, or MASTER_ADDR is a loopback address):
        unix socket FILE "<private dir>/rdzv-<MASTER_PORT>-<run id>", mode 0600, in a directory of mode 0700 owned by this
        user ($XDG_RUNTIME_DIR/fastmc or /tmp/fastmc-<uid>): only processes of the same user can connect (the abstract
        namespace has no permissions at all), and no TCP port collides with the launcher's own store
  otherwise: tcp://MASTER_ADDR:(MASTER_PORT + 1)

Who may join: every connection starts with a mutual HMAC-SHA256 challenge (rank 0 sends a nonce, the peer answers with
its (rank, world) hello signed with the shared token and its own nonce, rank 0 signs that back).  The token is
FASTMC_RDZV_TOKEN when the launcher exports one (do, across nodes), else derived from the launcher's run id and port --
which keeps strangers' stray connections out but is not a secret; a TCP rendezvous without FASTMC_RDZV_TOKEN says so once.
Messages carry only bytes of numpy arrays and JSON (fast_amd/sweep.py); nothing received is ever unpickled, and a length
prefix beyond FASTMC_RDZV_MAX_MSG (default 1 GiB) closes the connection.

The reference (ojdf/fast) is single-process; this has no counterpart there.
"""
import atexit
import hashlib
import hmac
import logging
import os
import socket
import stat
import struct
import time

import numpy as np


logger = logging.getLogger(__name__)
MAX_MSG = int(os.environ.get("FASTMC_RDZV_MAX_MSG", str(1 << 30)))


class RendezvousError(RuntimeError):
    pass


def _token():
    """Shared key of the hello handshake (bytes) and whether the launcher supplied it explicitly."""
    t = os.environ.get("FASTMC_RDZV_TOKEN")
    if t:
        return t.encode(), True
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "") or os.environ.get("SLURM_JOB_ID", "")
    return f"fastmc|{run_id}|{os.environ.get('MASTER_PORT', '29400')}|{os.getuid()}".encode(), False


def _private_dir():
    """A directory only this user can enter, for the unix socket file: $XDG_RUNTIME_DIR/fastmc, else /tmp/fastmc-<uid>."""
    base = os.environ.get("XDG_RUNTIME_DIR")
    d = os.path.join(base, "fastmc") if base and os.path.isdir(base) and os.access(base, os.W_OK) else \
        os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fastmc-{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RendezvousError(f"{d} must be a directory of mode 0700 owned by uid {os.getuid()} (found mode {oct(st.st_mode & 0o7777)}, uid {st.st_uid})")
    return d


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
        run_id = "".join(c if c.isalnum() or c in "-_." else "_" for c in run_id)[:40]
        return "unix", os.path.join(_private_dir(), f"rdzv-{port}-{run_id}")
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


def _recv(sock, limit=None):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > (MAX_MSG if limit is None else limit):
        raise RendezvousError(f"rendezvous message of {n} bytes exceeds the limit (FASTMC_RDZV_MAX_MSG)")
    return _recv_exact(sock, n)


def _sign(key, *parts):
    return hmac.new(key, b"|".join(parts), hashlib.sha256).digest()


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
        # a unix endpoint with a '/' is a socket FILE (mode 0600 in a private directory); a bare name (explicit
        # FASTMC_RDZV=unix:name only) lives in the abstract namespace, which has no permissions: the handshake alone guards it
        self._sock_file = address if (kind == "unix" and "/" in address) else None
        addr = (address if self._sock_file else "\0" + address) if kind == "unix" else address
        self.endpoint = f"{kind}:{address}"
        self._key, explicit = _token()
        if self.world == 1:
            return
        if kind == "tcp" and not explicit and self.rank == 0:
            logger.warning("TCP rendezvous without FASTMC_RDZV_TOKEN: the hello handshake is keyed on the run id and port only; "
                           "export a secret FASTMC_RDZV_TOKEN to every rank on networks you do not trust")
        if self.rank == 0:
            ls = socket.socket(fam, socket.SOCK_STREAM)
            if kind == "tcp":
                ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                if self._sock_file:
                    try:
                        os.unlink(self._sock_file)          # a stale file of an earlier run with the same port / run id
                    except FileNotFoundError:
                        pass
                    old = os.umask(0o177)
                    try:
                        ls.bind(addr)
                    finally:
                        os.umask(old)
                else:
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
                    c.settimeout(min(self.timeout, 20.0))
                    if kind == "tcp":
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    # mutual challenge: a peer that cannot sign our nonce with the shared token is dropped, not fatal
                    try:
                        nonce = os.urandom(16)
                        _send(c, nonce)
                        hello = _recv(c, limit=256)
                        if len(hello) != 8 + 16 + 32:
                            raise RendezvousError("malformed hello")
                        (r, w) = struct.unpack("<ii", hello[:8])
                        peer_nonce, mac = hello[8:24], hello[24:]
                        if not hmac.compare_digest(mac, _sign(self._key, b"hello", nonce, hello[:8], peer_nonce)):
                            raise RendezvousError("hello not signed with this run's token")
                        if w != self.world or not 0 < r < self.world or r in peers:
                            raise RendezvousError(f"unexpected hello (rank {r} of {w})")
                        _send(c, _sign(self._key, b"welcome", peer_nonce, hello[:8]))
                    except (RendezvousError, OSError, struct.error) as e:
                        logger.warning(f"rendezvous {self.endpoint}: connection refused ({e})")
                        c.close()
                        continue
                    c.settimeout(self.timeout)
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
            try:
                nonce = _recv(s, limit=64)
                me, mine = struct.pack("<ii", self.rank, self.world), os.urandom(16)
                _send(s, me + mine + _sign(self._key, b"hello", nonce, me, mine))
                welcome = _recv(s, limit=64)
            except (OSError, RendezvousError) as e:
                raise RendezvousError(f"rank {self.rank}: handshake with rank 0 at {self.endpoint} failed ({e}): do all ranks share "
                                      "FASTMC_RDZV_TOKEN / the launcher's run id and MASTER_PORT?")
            if not hmac.compare_digest(welcome, _sign(self._key, b"welcome", mine, me)):
                raise RendezvousError(f"rank {self.rank}: the process listening at {self.endpoint} is not rank 0 of this run")
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
        if self.rank == 0 and getattr(self, "_sock_file", None):
            try:
                os.unlink(self._sock_file)
            except OSError:
                pass
            self._sock_file = None


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
