"""CPU: the N > 1 path without torch -- fast_amd's own rendezvous (plain sockets), the collective choice of the
exchange transport, sharding, gather and histogram reduce, with 2 and 3 ranks started as plain processes; and the
one-process / N-threads driver (fast_amd/multi.py) on stand-in handles."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _launch(world, port, extra_env=None):
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", FASTMC_RDZV_TIMEOUT="60")
        env.pop("FASTMC_RDZV", None)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rdzv_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o[-2000:] + e[-3000:]
    return outs[0][0]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_over_unix_socket_rendezvous(world):
    out = _launch(world, 29610 + world)
    assert f"RDZV OK {world} unix host" in out


def test_ranks_over_tcp_rendezvous_and_rccl_disabled():
    out = _launch(2, 29620, {"FASTMC_RDZV": "tcp://127.0.0.1:29633", "FASTMC_DISABLE_RCCL": "1"})
    assert "RDZV OK 2 tcp host" in out


def test_single_process_is_not_a_rendezvous():
    from fast_amd import rendezvous
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK") if k in os.environ}
    try:
        assert rendezvous.from_env() is None
    finally:
        os.environ.update(env)


def test_device_group_threads_assemble_like_one_device():
    """fast_amd.multi.DeviceGroup on stand-in handles: ragged contiguous pieces, host log-amplitudes cut per shard,
    complex results, histogram sum -- the vector of an unsharded run, bit for bit."""
    from fast_amd import multi

    class Fake:
        def __init__(self, dev):
            self.device = dev
            self.calls = []

        def run(self, seed, real0, n, la, lvar, coherent):
            self.calls.append((real0, n))
            g = np.arange(real0, real0 + n, dtype=float)
            re, im = np.sin(g * seed) + 2, np.cos(g * seed) + 2
            if la is not None:
                re, im = re * np.exp(la[:n]), im * np.exp(la[n:])
            out = np.concatenate([re, im])
            self._last = out
            return out * (1 + 1j) if coherent else out

        def histogram(self, lo, hi, nb):
            return np.histogram(10 * np.log10(self._last), bins=nb, range=(lo, hi))[0]

        def close(self):
            pass

    for ndev, n_real in ((1, 7), (2, 7), (3, 10), (4, 3)):
        grp = multi.DeviceGroup(8, 4, "f64", list(range(ndev)), factory=Fake)
        one = Fake(0)
        la = np.linspace(-0.2, 0.2, 2 * n_real)
        for logamp in (None, la):
            for coh in (False, True):
                want = one.run(3, 5, n_real, logamp, 0.0, coh)
                got = grp.run(3, 5, n_real, logamp, 0.0, coh)
                assert np.array_equal(got, want)
        assert sum(n for h in grp.handles for _, n in h.calls[:1]) == n_real
        assert grp.exchange == ("none" if ndev == 1 else "host")
        want = one.run(3, 5, n_real, None, 0.0, False)
        grp.run(3, 5, n_real, None, 0.0, False, hist_range=(0.0, 6.0, 12))
        assert np.array_equal(grp.last_hist, np.histogram(10 * np.log10(want), bins=12, range=(0.0, 6.0))[0])


def test_rendezvous_refuses_strangers_and_oversized_messages(tmp_path, monkeypatch):
    """The hello is a mutual HMAC challenge: a connection that cannot sign rank 0's nonce with the run's token is dropped
    (rank 0 keeps waiting for the real rank), a rank with another token is told so, and a length prefix beyond the limit is an
    error instead of an allocation.  The one-node endpoint is a socket FILE of mode 0600 in a private directory."""
    import socket
    import stat
    import struct
    import threading
    from fast_amd import rendezvous as rz
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.delenv("XDG_RUNTIME_DIR", raising=False)
    monkeypatch.delenv("FASTMC_RDZV", raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29644")
    monkeypatch.setenv("FASTMC_RDZV_TOKEN", "s3cret")
    kind, path = rz._endpoint(2)
    assert kind == "unix" and path.startswith(str(tmp_path)) and stat.S_IMODE(os.lstat(os.path.dirname(path)).st_mode) == 0o700
    box = {}

    def rank0():
        try:
            box["r0"] = rz.Rendezvous(0, 2, kind, path, timeout=20)
            box["got"] = box["r0"].exchange(b"zero")
        except Exception as e:
            box["err"] = e
    th = threading.Thread(target=rank0)
    th.start()
    # a stranger: connects, answers the nonce with garbage
    for _ in range(200):
        try:
            s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            s.connect(path)
            break
        except OSError:
            s.close()
            import time
            time.sleep(0.02)
    assert stat.S_IMODE(os.lstat(path).st_mode) == 0o600
    rz._recv(s)
    rz._send(s, struct.pack("<ii", 1, 2) + b"\0" * 48)
    assert s.recv(1) == b""                      # dropped
    s.close()
    # a rank of another run (wrong token)
    monkeypatch.setenv("FASTMC_RDZV_TOKEN", "other")
    with pytest.raises(rz.RendezvousError, match="handshake|not rank 0"):
        rz.Rendezvous(1, 2, kind, path, timeout=5)
    # the real rank 1
    monkeypatch.setenv("FASTMC_RDZV_TOKEN", "s3cret")
    r1 = rz.Rendezvous(1, 2, kind, path, timeout=10)
    assert r1.exchange(b"one") == [b"zero", b"one"]
    th.join(20)
    assert "err" not in box and box["got"] == [b"zero", b"one"]
    # an absurd length prefix
    a, b = socket.socketpair()
    a.sendall(struct.pack("<Q", 1 << 50))
    with pytest.raises(rz.RendezvousError, match="exceeds the limit"):
        rz._recv(b)
    box["r0"].close()
    r1.close()
    assert not os.path.exists(path)


def test_sweep_records_travel_as_json_not_pickle():
    from fast_amd import sweep
    recs = [{"index": 1, "zenith": 12.5, "r": np.arange(6, dtype=float).reshape(2, 3), "n": np.int64(3), "name": "x"}]
    blob = sweep._encode(recs)
    assert blob.lstrip().startswith(b"[") and b"pickle" not in blob
    back = sweep._decode(blob)
    assert back[0]["index"] == 1 and back[0]["n"] == 3 and np.array_equal(back[0]["r"], recs[0]["r"]) and back[0]["r"].dtype == float
    with pytest.raises(Exception):
        sweep._decode(b'{"not": "a list"}')
    with pytest.raises(TypeError):
        sweep._encode([{"index": 0, "bad": np.array([object()])}])
    src = open(os.path.join(ROOT, "fast_amd", "sweep.py")).read() + open(os.path.join(ROOT, "fast_amd", "rendezvous.py")).read()
    assert "import pickle" not in src and "pickle.loads" not in src
