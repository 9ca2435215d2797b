"""One process drives N GPUs: N handles, one thread each (`GPU_DEVICES: [0, 1, ...]` on `Fast`, `bench.py --gpus N`).

SURVEY 8(e): iterations are independent (fast/fast.py:130-134, 589-605), the spectrum and pupil weights are small
(8-32 MB) and replicated by one host copy per device, and the only exchange is at the end of a run.  Every handle owns
a HIP stream and ctypes releases the GIL during library calls, so N Python threads keep N devices busy; the exchange is
RCCL over xGMI when `ncclCommInitAll` gives this process a communicator clique (fastmc_comm_init_all: grouped
ncclAllGather of the powers, ncclAllReduce of the dB histogram on the device buffers, issued from one thread), else the
host concatenation of the vectors `fastmc_run` already returned.  Which one ran is reported in `exchange`.

A sharded step is ONE synchronisation per device: every handle's kernels are enqueued without waiting
(`fastmc_run_async`), the grouped collectives follow on the same streams and only the gathered result is copied back.
The exchange runs under a deadline (`FASTMC_EXCHANGE_TIMEOUT`, seconds, default 120): when it does not return in time,
or fails, the clique is aborted (`fastmc_comm_abort` = ncclCommAbort, which never waits for peers), every device's own
vector is fetched (`fastmc_wait`) and concatenated on the host, and the group stays on the host exchange from then on
(`exchange` says why).  `FASTMC_TEST_STALL_GATHER=1` injects exactly that fault (tests/test_gpu_dist.py).

Nothing here imports torch or needs a launcher.  The reference is single-threaded; no counterpart.
"""
import logging
import os
import threading
import time

import numpy as np

from . import _lib, dist

logger = logging.getLogger(__name__)


def run_threads(fns, pool=None):
    """Run the callables concurrently, one thread each (on `pool`, a ThreadPoolExecutor with at least len(fns) workers,
    when given: a sharded run is called once per step and should not pay thread start-up each time); return their results
    in order; re-raise the first error."""
    if len(fns) == 1:
        return [fns[0]()]
    if pool is not None:
        futs = [pool.submit(f) for f in fns]
        res, err = [], None
        for f in futs:
            try:
                res.append(f.result())
            except BaseException as e:   # wait for all of them, then re-raise the first
                res.append(None)
                err = err or e
        if err is not None:
            raise err
        return res
    res = [None] * len(fns)
    err = [None] * len(fns)

    def work(i):
        try:
            res[i] = fns[i]()
        except BaseException as e:       # re-raised in the caller's thread
            err[i] = e
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return res


call_with_deadline, exchange_timeout = dist.call_with_deadline, dist.exchange_timeout


def _virtual_ranks():
    """The tests' stand-in for librccl.so (FASTMC_RCCL_LIB) takes several ranks on ONE device when FASTMC_TEST_VIRTUAL_RANKS=1
    (tests/stubs/fake_rccl.cpp; fastmc.hip: cslot): the group / stream / ordering code of the RCCL path then runs with a world of
    2 ... 8 on the single GPU of a test box.  Never with the real library."""
    return bool(os.environ.get("FASTMC_RCCL_LIB")) and os.environ.get("FASTMC_TEST_VIRTUAL_RANKS", "0") not in ("", "0")


class DeviceGroup:
    """N handles of one (N_grid, Np, precision) problem on N devices.

    handles[i] lives on devices[i]; `run` cuts the realisation range into contiguous pieces (sizes differing by at most
    one), runs them concurrently and returns the vector an unsharded `Handle.run` would return, bit for bit.
    `exchange`: 'rccl' or 'host (reason)'.  `factory(device) -> handle-like` lets the CPU tests stand in for the GPU."""

    def __init__(self, N, Np, precision="f64", devices=(0,), exchange="auto", rccl_timeout=None, factory=None):
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise Exception("GPU_DEVICES must name at least one device")
        make = factory or (lambda d: _lib.Handle(N, Np, precision, d))
        self._pool = None
        if len(self.devices) > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="fastmc-dev")
        self.handles = run_threads([(lambda d=d: make(d)) for d in self.devices], self._pool)
        self.world = len(self.handles)
        self.exchange = "none" if self.world == 1 else "host"
        self._rccl = False
        self.rccl_ranks = 0            # world size the communicators report (0: no clique)
        self.degraded = None           # why an RCCL clique was given up mid-run, if it was
        self.last_exchange = "none"
        self.last_exchange_ms = []     # per device: HIP-event time of the collectives of the last step (RCCL path)
        self.last_exchange_wall_ms = 0.0
        self._stall_test = os.environ.get("FASTMC_TEST_STALL_GATHER", "0") == "1"
        if self.world > 1 and exchange in ("auto", "rccl") and factory is None:
            if self._stall_test:
                # fault injection: behave as if the clique were up; the exchange entry point blocks until it is aborted
                self._rccl, self.exchange, self.rccl_ranks = True, "rccl", self.world
            else:
                self._try_rccl(rccl_timeout)
            if exchange == "rccl" and not self._rccl:
                raise _lib.FastMCError(f"RCCL exchange requested but unavailable: {self.exchange}")

    def _try_rccl(self, timeout):
        timeout = float(os.environ.get("FASTMC_RCCL_TIMEOUT", "90")) if timeout is None else timeout
        if len(set(self.devices)) < self.world and not _virtual_ranks():
            self.exchange = "host (several handles share a device: RCCL needs one device per rank)"
            return
        # the communicators belong to the devices and outlive the handles: a clique left by an earlier group of the
        # same devices in the same order is reused
        try:
            if all(h.comm_world() == (self.world, i) for i, h in enumerate(self.handles)):
                self._rccl, self.exchange, self.rccl_ranks = True, "rccl", self.world
                return
        except Exception:
            pass
        ok, val = call_with_deadline(lambda: _lib.comm_init_all(self.handles), timeout)
        if ok:
            self._rccl, self.exchange = True, "rccl"
            self.rccl_ranks = self.handles[0].comm_world()[0]
        else:
            # a clique that is still being built (or half built) is aborted, never destroyed: ncclCommAbort does not wait
            # for peers, and an init that returns later finds the abort and drops its communicators (fastmc_comm_init_all)
            for h in self.handles:
                try:
                    h.comm_abort()
                except Exception:
                    pass
            self.exchange = f"host ({val})"
            logger.warning(f"RCCL exchange unavailable ({self.exchange}); results are concatenated on the host")

    # ---- broadcast of the problem (one host copy per device)
    def each(self, fn):
        return run_threads([(lambda h=h, i=i: fn(h, i)) for i, h in enumerate(self.handles)], self._pool)

    def set_spectrum(self, powerspec, df):
        self.each(lambda h, i: h.set_spectrum(powerspec, df))

    def set_pupil(self, W, crop_lo, dx):
        self.each(lambda h, i: h.set_pupil(W, crop_lo, dx))

    def set_subharm(self, *a):
        self.each(lambda h, i: h.set_subharm(*a))

    def set_batch(self, batch):
        self.each(lambda h, i: h.set_batch(batch))

    # ---- the sharded run
    def run(self, seed, real0, n_real, logamp=None, logamp_var=0.0, coherent=False, hist_range=None):
        """Realisations [real0, real0 + n_real) over the devices; returns the full vector in `Handle.run`'s order
        ([Re-screen results | Im-screen results]).  With hist_range = (lo_db, hi_db, nbins) also leaves the global dB
        histogram in `self.last_hist` (device all-reduce when RCCL is up)."""
        ranges = dist.shard_ranges(n_real, self.world)
        la = None if logamp is None else np.ascontiguousarray(logamp, dtype=np.float64)
        equal = len({n for _, n in ranges}) == 1 and ranges[0][1] > 0
        self.last_hist = None
        self.last_exchange_ms, self.last_exchange_wall_ms = [], 0.0
        if self._rccl and equal and la is None:
            return self._run_rccl(seed, real0, ranges, logamp_var, coherent, hist_range)

        def piece(h, i):
            r0, n = ranges[i]
            if n == 0:
                return np.empty(0, dtype=np.complex128 if coherent else np.float64)
            lai = None
            if la is not None:        # [Re block | Im block] of the whole range -> this shard's two blocks
                lai = np.concatenate([la[r0:r0 + n], la[n_real + r0:n_real + r0 + n]])
            return h.run(seed, real0 + r0, n, lai, logamp_var, coherent)
        parts = self.each(piece)
        t0 = time.perf_counter()
        self.last_exchange = "host" if self.world > 1 else "none"
        if hist_range is not None:
            hs = self.each(lambda h, i: h.histogram(*hist_range) if ranges[i][1] else 0)
            self.last_hist = np.sum([x for x in hs if not np.isscalar(x)], axis=0)
        out = dist.assemble(parts, complex_out=coherent)
        self.last_exchange_wall_ms = (time.perf_counter() - t0) * 1e3
        if self.degraded is not None:
            dist.mark_clean_exit()      # a group that gave its clique up finished a step on the host path (dist: exit hooks)
        return out

    def _run_rccl(self, seed, real0, ranges, logamp_var, coherent, hist_range):
        """One synchronisation per device: kernels enqueued on every stream, grouped all-gather (+ histogram all-reduce) behind
        them, result copied from rank 0.  Under a deadline; on a miss the clique is aborted and the step finishes on the host."""
        n = ranges[0][1]
        self.each(lambda h, i: h.run_async(seed, real0 + ranges[i][0], n, logamp_var, coherent))
        nval = 2 * n * (2 if coherent else 1)
        t0 = time.perf_counter()
        ok, val = call_with_deadline(lambda: _lib.comm_gather_all(self.handles, nval, hist_range), exchange_timeout())
        if ok:
            allp, hist = val
            allp = allp.reshape(self.world, nval)
            parts = [allp[r].view(np.complex128) if coherent else allp[r] for r in range(self.world)]
            self.last_hist = hist
            self.last_exchange = "rccl"
            self.last_exchange_ms = [h.last_exchange_ms() for h in self.handles]
            self.last_exchange_wall_ms = (time.perf_counter() - t0) * 1e3     # includes the wait for the kernels
            return dist.assemble(parts, complex_out=coherent)
        self._degrade(val)
        # every device's own vector is still resident; the streams carry the aborted collectives, so the fetch has a deadline
        ok, parts = call_with_deadline(lambda: self.each(lambda h, i: h.wait()), dist.post_abort_timeout())
        if not ok:
            raise dist.ExchangeStuck(f"the devices' results could not be fetched after the RCCL exchange was aborted: {parts}")
        if hist_range is not None:
            ok, hs = call_with_deadline(lambda: self.each(lambda h, i: h.histogram(*hist_range)), dist.post_abort_timeout())
            if not ok:
                raise dist.ExchangeStuck(f"histogram after the aborted RCCL exchange: {hs}")
            self.last_hist = np.sum(hs, axis=0)
        self.last_exchange = "host"
        self.last_exchange_wall_ms = (time.perf_counter() - t0) * 1e3
        dist.mark_clean_exit()          # the step finished on the host path after its device exchange was given up
        return dist.assemble(parts, complex_out=coherent)

    # ---- the same steps with TWO in flight: the device never waits for the host between steps
    def run_pipelined(self, seed, steps, logamp_var=0.0, coherent=False, hist_range=None):
        """Generator over `steps` = [(real0, n_real), ...]: yields (vector, histogram | None) of each step, in order, exactly
        what `run` returns for it -- but step i + 1's kernels (and its exchange) are enqueued BEFORE step i's results are
        waited for (fastmc_run_queued / fastmc_comm_gather_all_queued / fastmc_queue_wait: two slots per handle), so the
        per-step host work (launches, the exchange's host side, result copies, Python) is hidden behind the device's work.
        The exchange of every step runs under the same deadline as in `run`; a step whose device exchange gives no result is
        recomputed -- the generator is keyed on the realisation index, so that is the same vector -- on the host path, which the
        group then keeps."""
        steps = [(int(r0), int(n)) for r0, n in steps]
        self.last_hist = None
        nbins = hist_range[2] if hist_range is not None else 0

        def shards(n_real):
            return dist.shard_ranges(n_real, self.world)

        def enqueue(i):
            r0, n_real = steps[i]
            rg = shards(n_real)
            slot = i & 1
            equal = len({n for _, n in rg}) == 1 and rg[0][1] > 0
            use_rccl = self._rccl and equal and self.world > 1
            # (launches only: a few tens of microseconds per handle, from this thread -- they are off the critical path)
            for h, (s0, n) in zip(self.handles, rg):
                if n:
                    h.run_queued(seed, r0 + s0, n, logamp_var, coherent, slot, fetch=not use_rccl)
            if use_rccl:
                _lib.comm_gather_all_queued(self.handles, 2 * rg[0][1] * (2 if coherent else 1), hist_range, True, slot)
            elif hist_range is not None:
                for h, (s0, n) in zip(self.handles, rg):
                    if n:
                        h.histogram_queued(*hist_range, slot=slot)
            return use_rccl

        def collect(i, used_rccl):
            r0, n_real = steps[i]
            rg = shards(n_real)
            slot = i & 1
            if used_rccl:
                nval = 2 * rg[0][1] * (2 if coherent else 1)
                allp, hist = self.handles[0].queue_wait(slot, nval * self.world, nbins)
                for h in self.handles[1:]:
                    h.queue_wait(slot)
                allp = allp.reshape(self.world, nval)
                parts = [allp[r].view(np.complex128) if coherent else allp[r] for r in range(self.world)]
                self.last_exchange = "rccl"
                self.last_exchange_ms = [h.last_exchange_ms() for h in self.handles]
                return dist.assemble(parts, complex_out=coherent), hist
            parts, hist = [], None
            for h, (s0, n) in zip(self.handles, rg):
                if not n:
                    parts.append(np.empty(0, dtype=np.complex128 if coherent else np.float64))
                    continue
                v, hh = h.queue_wait(slot, 2 * n * (2 if coherent else 1), nbins)
                parts.append(v.view(np.complex128) if coherent else v)
                if hh is not None:
                    hist = hh if hist is None else hist + hh
            self.last_exchange = "host" if self.world > 1 else "none"
            return dist.assemble(parts, complex_out=coherent), hist

        def drain():
            for h in self.handles:
                for slot in (0, 1):
                    try:
                        h.queue_wait(slot)
                    except Exception:
                        pass

        mode = {}
        done = False
        try:
            if steps:
                mode[0] = enqueue(0)
            i = 0
            while i < len(steps):
                if i + 1 < len(steps) and (i + 1) not in mode:
                    mode[i + 1] = enqueue(i + 1)
                t0 = time.perf_counter()
                if mode[i]:
                    ok, val = call_with_deadline(lambda: collect(i, True), exchange_timeout())
                else:
                    ok, val = True, collect(i, False)
                self.last_exchange_wall_ms = (time.perf_counter() - t0) * 1e3
                if not ok:
                    # the device exchange gave no result: abort, wait for whatever is still inside the library, empty both slots
                    # and redo this step and the queued one on the host path (same realisations, same numbers)
                    self._degrade(val)
                    ok2, why2 = call_with_deadline(drain, dist.post_abort_timeout())
                    if not ok2:
                        raise dist.ExchangeStuck(f"queued steps did not drain after the RCCL exchange was aborted: {why2}")
                    mode = {i: enqueue(i)}
                    continue
                self.last_hist = val[1]
                mode.pop(i, None)
                if self.degraded is not None:
                    dist.mark_clean_exit()
                yield val
                i += 1
            done = True
        finally:
            # A consumer that stops early (break, an exception in its loop body, the generator collected) leaves step i + 1
            # enqueued with its slot busy -- and on the RCCL path a collective on the stream: the next pass over these handles
            # would find "this slot is still in flight".  Empty both slots of every handle, under the post-abort deadline.
            if not done and mode:
                ok3, why3 = call_with_deadline(drain, dist.post_abort_timeout())
                if not ok3:
                    logger.warning(f"queued steps did not drain when the pipelined pass was abandoned: {why3}")

    def _degrade(self, why):
        """Give the clique up for good: abort every communicator (wakes a blocked exchange) and take the host path."""
        self._rccl = False
        self.degraded = why
        self.exchange = f"host (RCCL exchange given up: {why})"
        logger.warning(f"RCCL exchange gave no result ({why}); communicators aborted, results are concatenated on the host from now on")
        for h in self.handles:
            try:
                h.comm_abort()
            except Exception as e:
                logger.warning(f"ncclCommAbort on device {h.device}: {e}")
        # the exchange that missed its deadline may still be inside the library on these handles (it holds their locks there):
        # wait for it, with a bound, before anything else touches them
        if not dist.join_left_behind(dist.post_abort_timeout()):
            raise dist.ExchangeStuck(f"the aborted RCCL exchange did not return within {dist.post_abort_timeout():g} s ({why})")

    def last_timing(self):
        return [h.last_timing() for h in self.handles]

    def __del__(self):
        try:
            if self._pool is not None:
                self._pool.shutdown(wait=False)
        except Exception:
            pass

    def close(self, destroy_comm=False):
        if self._rccl and destroy_comm:
            for h in self.handles:
                try:
                    h.comm_destroy()
                except Exception:
                    pass
        self._rccl = False
        for h in self.handles:
            h.close()
        if self._pool is not None:
            self._pool.shutdown(wait=False)
            self._pool = None
