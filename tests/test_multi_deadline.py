"""CPU: the deadline / abort / host fall-back around the per-step exchange of fast_amd/multi.py (one process, N devices),
with stand-in handles -- what a hung `fastmc_comm_gather_all` looks like to DeviceGroup.  The real thing runs under
`-m gpu` (tests/test_gpu_dist.py: FASTMC_TEST_STALL_GATHER)."""
import threading
import time

import numpy as np

from fast_amd import _lib, dist, multi


class FakeHandle:
    def __init__(self, device):
        self.device, self.aborted, self.waited = device, 0, 0

    def _values(self, real0, n):
        g = np.arange(real0, real0 + n, dtype=float)
        return np.concatenate([g + 0.25, g + 0.75])          # [Re-screen results | Im-screen results]

    def run_async(self, seed, real0, n, logamp_var, coherent):
        self._local = self._values(real0, n)

    def wait(self):
        self.waited += 1
        return self._local

    def run(self, seed, real0, n, logamp, logamp_var, coherent):
        return self._values(real0, n)

    def histogram(self, lo, hi, nbins):
        h = np.zeros(nbins + 2, dtype=np.int64)
        h[0] = self._local.size
        return h

    def comm_abort(self):
        self.aborted += 1
        FakeHandle.wake.set()

    # two steps in flight
    def run_queued(self, seed, real0, n, logamp_var, coherent, slot, fetch=True):
        q = self.__dict__.setdefault("_q", {})
        assert slot not in q, "slot busy"
        q[slot] = self._values(real0, n)
        self._local = q[slot]

    def histogram_queued(self, lo, hi, nbins, slot=0):
        h = np.zeros(nbins + 2, dtype=np.int64)
        h[0] = self._q[slot].size
        self.__dict__.setdefault("_qh", {})[slot] = h

    def queue_wait(self, slot, n_out=0, hist_bins=0):
        v = self._q.pop(slot)
        hh = self.__dict__.get("_qh", {}).pop(slot, None)
        return (v if n_out else None), (hh if hist_bins else None)

    def last_exchange_ms(self):
        return 0.5

    def last_timing(self):
        return {}

    def close(self):
        pass


def _group(n):
    FakeHandle.wake = threading.Event()
    grp = multi.DeviceGroup(64, 8, "f64", list(range(n)), factory=FakeHandle)
    grp._rccl, grp.exchange, grp.rccl_ranks = True, "rccl", n       # as after a successful ncclCommInitAll
    return grp


def test_exchange_that_never_answers_is_aborted_and_the_step_finishes_on_the_host(monkeypatch):
    grp = _group(4)
    monkeypatch.setenv("FASTMC_EXCHANGE_TIMEOUT", "0.5")

    def hung(handles, n_local, hist_range=None, powers=True):
        FakeHandle.wake.wait()                     # until some handle's comm_abort
        raise _lib.FastMCError("aborted")
    monkeypatch.setattr(_lib, "comm_gather_all", hung)
    t0 = time.perf_counter()
    out = grp.run(1, 100, 40, None, 0.0, False, hist_range=(-10.0, 10.0, 4))
    assert 0.4 < time.perf_counter() - t0 < 5.0
    want = np.concatenate([np.arange(100, 140) + 0.25, np.arange(100, 140) + 0.75])
    assert np.array_equal(out, want)
    assert grp.last_exchange == "host" and grp.exchange.startswith("host (RCCL exchange given up") and "no answer" in grp.degraded
    assert all(h.aborted == 1 and h.waited == 1 for h in grp.handles) and grp.last_hist[0] == 80
    time.sleep(0.1)
    assert not dist.stuck_threads()                # the abort woke the thread the deadline left behind
    # later steps never try the device exchange again
    monkeypatch.setattr(_lib, "comm_gather_all", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not be called")))
    out2 = grp.run(1, 0, 8, None, 0.0, False)
    assert np.array_equal(out2, np.concatenate([np.arange(8) + 0.25, np.arange(8) + 0.75])) and grp.last_exchange == "host"


def test_exchange_error_falls_back_and_success_is_taken(monkeypatch):
    grp = _group(2)
    monkeypatch.setattr(_lib, "comm_gather_all", lambda *a, **k: (_ for _ in ()).throw(_lib.FastMCError("libfastmc error -5: ncclAllGather: unhandled system error")))
    out = grp.run(1, 0, 10, None, 0.0, False)
    assert np.array_equal(out, np.concatenate([np.arange(10) + 0.25, np.arange(10) + 0.75]))
    assert "unhandled system error" in grp.exchange and all(h.aborted == 1 for h in grp.handles)

    grp = _group(2)

    def fine(handles, n_local, hist_range=None, powers=True):
        return np.concatenate([h._local for h in handles]), np.array([1, 2, 3])
    monkeypatch.setattr(_lib, "comm_gather_all", fine)
    out = grp.run(1, 0, 10, None, 0.0, False, hist_range=(-1.0, 1.0, 1))
    assert np.array_equal(out, np.concatenate([np.arange(10) + 0.25, np.arange(10) + 0.75]))
    assert grp.last_exchange == "rccl" and grp.last_exchange_ms == [0.5, 0.5] and list(grp.last_hist) == [1, 2, 3]
    assert all(h.aborted == 0 and h.waited == 0 for h in grp.handles)
    # unequal shards (or host-supplied log-amplitudes) cannot use the all-gather: host path without touching the clique
    out = grp.run(1, 0, 11, None, 0.0, False)
    assert out.size == 22 and grp.last_exchange == "host" and grp.exchange == "rccl"


# ---- exit status with a thread left inside the library (fast_amd/dist.py: _exit_hook)
import os
import subprocess
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_STUCK = """
import sys, threading
sys.path.insert(0, {root!r})
from fast_amd import dist
ok, why = dist.call_with_deadline(threading.Event().wait, 0.2)      # a thread that never comes back
assert not ok and dist.stuck_threads()
{tail}
"""


def _rc(tail):
    r = subprocess.run([sys.executable, "-c", _STUCK.format(root=_ROOT, tail=tail)], capture_output=True, text=True, timeout=60)
    return r.returncode, r.stderr


def test_a_stuck_thread_never_turns_an_error_exit_into_success():
    rc, err = _rc("raise RuntimeError('the run died after an exchange timed out')")
    assert rc == 1 and "the run died" in err
    assert _rc("sys.exit(3)")[0] == 3
    assert _rc("raise SystemExit(4)")[0] in (4, 70)   # not via sys.exit: no hook sees it -- "nobody said the run finished", not 0
    assert _rc("try:\n    sys.exit(5)\nexcept SystemExit:\n    pass\ndist.mark_clean_exit()")[0] == 0     # a caught exit is history
    assert _rc("pass")[0] == 70                       # nobody said the run finished: EX_SOFTWARE, not 0
    assert _rc("dist.mark_clean_exit()")[0] == 0      # the work is done and reported: the stuck thread is abandoned quietly
    assert _rc("dist.mark_clean_exit(); raise ValueError('late failure')")[0] == 1
    # a failure path that ends in a raised SystemExit after a step marked the exit clean (bench.py --require-rccl did)
    # came out as 0.  No hook Python offers sees a top-level `raise SystemExit(n)`, so such a path records its status first:
    assert _rc("dist.mark_clean_exit(); dist.mark_exit(4); raise SystemExit(4)")[0] == 4
    assert _rc("dist.mark_clean_exit(); dist.mark_exit('degraded to the host exchange'); raise SystemExit('degraded')")[0] == 1
    assert _rc("dist.mark_clean_exit(); dist.mark_exit(3); sys.exit(3)")[0] == 3
    assert _rc("dist.mark_exit(3); dist.mark_clean_exit()")[0] == 0       # ... and a later clean mark is the last word


def test_exit_hooks_exist_only_while_a_thread_is_left_behind():
    """importing fast_amd patches nothing; the hooks appear when a deadline leaves a thread behind and go away when the
    thread has come back (join_left_behind)."""
    code = """
import sys, threading
hook, ex = sys.excepthook, sys.exit
sys.path.insert(0, {root!r})
import fast_amd
from fast_amd import dist
assert sys.excepthook is hook and sys.exit is ex
gate = threading.Event()
ok, why = dist.call_with_deadline(gate.wait, 0.2)
assert not ok and dist.stuck_threads()
assert sys.excepthook is not hook and sys.exit is not ex
try:
    sys.exit(9)
except SystemExit as e:
    assert e.code == 9
gate.set()
assert dist.join_left_behind(5.0) and not dist.stuck_threads()
assert sys.excepthook is hook and sys.exit is ex
"""
    r = subprocess.run([sys.executable, "-c", code.format(root=_ROOT)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr


def test_left_behind_exchange_is_waited_for_with_a_bound(monkeypatch):
    """After the abort the step does not touch the handles before the exchange it gave up on has left them -- and if that
    never happens it fails (ExchangeStuck) instead of hanging or running on."""
    import pytest
    grp = _group(2)
    monkeypatch.setenv("FASTMC_EXCHANGE_TIMEOUT", "0.3")
    monkeypatch.setenv("FASTMC_POST_ABORT_TIMEOUT", "0.5")
    never = threading.Event()

    def deaf(handles, n_local, hist_range=None, powers=True):
        never.wait()                               # ignores the abort: still inside the library
    monkeypatch.setattr(_lib, "comm_gather_all", deaf)
    t0 = time.perf_counter()
    with pytest.raises(dist.ExchangeStuck):
        grp.run(1, 0, 8, None, 0.0, False)
    assert 0.7 < time.perf_counter() - t0 < 5.0
    assert all(h.aborted == 1 and h.waited == 0 for h in grp.handles)       # the handles were not touched
    never.set()
    time.sleep(0.1)
    dist._LEFT_BEHIND[:] = [t for t in dist._LEFT_BEHIND if t.is_alive()]


def test_pipelined_steps_equal_the_blocking_ones_and_survive_a_dead_exchange(monkeypatch):
    """DeviceGroup.run_pipelined: (1) host exchange, unequal shards, histograms; (2) with an RCCL clique whose queued exchange
    never answers: the deadline passes, the clique is aborted, both slots are drained and the SAME steps come back through the
    host path; later steps stay there."""
    steps = [(0, 10), (10, 11), (21, 10), (31, 4)]
    want = [np.concatenate([np.arange(r0, r0 + n) + 0.25, np.arange(r0, r0 + n) + 0.75]) for r0, n in steps]
    grp = _group(3)
    grp._rccl = False
    got = list(grp.run_pipelined(1, steps, 0.0, False, (-10.0, 10.0, 4)))
    for (v, hh), w, (_, n) in zip(got, want, steps):
        assert np.array_equal(v, w) and hh[0] == 2 * n
    assert all(not h._q for h in grp.handles)

    grp = _group(2)
    monkeypatch.setenv("FASTMC_EXCHANGE_TIMEOUT", "0.3")
    monkeypatch.setenv("FASTMC_POST_ABORT_TIMEOUT", "2")
    calls = {"n": 0}

    def gather_all_queued(handles, n_local, hist_range=None, powers=True, slot=0):
        calls["n"] += 1

    monkeypatch.setattr(_lib, "comm_gather_all_queued", gather_all_queued)
    real_wait = FakeHandle.queue_wait

    def wait_that_hangs_on_rccl_steps(self, slot, n_out=0, hist_bins=0):
        if grp._rccl and self is grp.handles[0] and n_out:       # the gathered landing buffer never arrives
            FakeHandle.wake.wait()
            raise _lib.FastMCError("aborted")
        return real_wait(self, slot, n_out, hist_bins)
    monkeypatch.setattr(FakeHandle, "queue_wait", wait_that_hangs_on_rccl_steps)
    even = [(0, 10), (10, 10), (20, 10)]
    want = [np.concatenate([np.arange(r0, r0 + n) + 0.25, np.arange(r0, r0 + n) + 0.75]) for r0, n in even]
    t0 = time.perf_counter()
    got = list(grp.run_pipelined(1, even, 0.0, False, None))
    assert 0.25 < time.perf_counter() - t0 < 10
    assert all(np.array_equal(v, w) for (v, _), w in zip(got, want))
    assert calls["n"] == 2                                        # steps 0 and 1 were queued on the clique before it was given up
    assert grp.exchange.startswith("host (RCCL exchange given up") and grp.last_exchange == "host"
    assert all(h.aborted == 1 for h in grp.handles)
    time.sleep(0.1)
    assert not dist.stuck_threads()
