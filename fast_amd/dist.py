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
import atexit
import logging
import os
import sys
import threading
import time

import numpy as np

logger = logging.getLogger(__name__)

# ---- deadlines around calls that may block inside RCCL
_LEFT_BEHIND = []          # daemon threads a deadline gave up on; normally they return once their communicator is aborted


def call_with_deadline(fn, timeout):
    """fn() on a daemon thread; (True, value) when it returned in time, (False, reason) when it raised or is still running
    after `timeout` seconds.  A thread that is still running is left behind: the caller aborts what it is blocked in
    (fastmc_comm_abort never waits for peers), which normally lets it return."""
    box = {}

    def work():
        try:
            box["value"] = fn()
        except BaseException as e:
            box["err"] = f"{type(e).__name__}: {e}"
    _EXIT["code"] = None          # an exit code recorded earlier did not end the process (see the exit hooks below)
    th = threading.Thread(target=work, daemon=True, name="fastmc-deadline")
    th.start()
    th.join(timeout)
    if th.is_alive():
        _LEFT_BEHIND.append(th)
        _EXIT["clean"] = False
        _install_exit_hooks()     # from here on the process may have to leave through os._exit: record how it meant to exit
        return False, f"no answer within {timeout:g} s"
    if "err" in box:
        return False, box["err"]
    return True, box["value"]


def exchange_timeout():
    """Deadline of one result exchange in seconds (it starts when the exchange is issued, i.e. it covers the wait for the
    step's kernels too): FASTMC_EXCHANGE_TIMEOUT, default 120."""
    return float(os.environ.get("FASTMC_EXCHANGE_TIMEOUT", "120"))


def stuck_threads():
    """Deadline threads that never came back (still inside RCCL after their communicator was aborted)."""
    return [t for t in _LEFT_BEHIND if t.is_alive()]


def join_left_behind(timeout):
    """After the abort: give the threads a deadline gave up on `timeout` seconds to leave the library (ncclCommAbort wakes a
    blocked exchange; the handle's lock inside libfastmc.so keeps them and the caller apart meanwhile).  True when none is left."""
    t_end = time.monotonic() + timeout
    for th in list(_LEFT_BEHIND):
        th.join(max(0.0, t_end - time.monotonic()))
    if stuck_threads():
        return False
    del _LEFT_BEHIND[:]
    _remove_exit_hooks()          # nobody is left inside the library: the interpreter can end the ordinary way again
    return True


def post_abort_timeout():
    """How long a step waits, after the communicators were aborted, for the exchange it gave up on to leave the handle and
    for the device's own vector (FASTMC_POST_ABORT_TIMEOUT, default 30 s): past it the step fails instead of hanging."""
    return float(os.environ.get("FASTMC_POST_ABORT_TIMEOUT", "30"))


class ExchangeStuck(RuntimeError):
    """An aborted exchange did not let go of its handle / stream in time: the step cannot finish on the host either."""


# ---- exit status when a thread is stuck inside RCCL at interpreter exit
# Interpreter / runtime teardown would wait for ever for such a thread, so the exit hook leaves through os._exit -- with the
# status the process was going to exit with.  Nothing is patched at import: the hooks that record that status (sys.excepthook for
# an uncaught exception, sys.exit for an exit code) are installed when the FIRST thread is left behind by a
# deadline and removed again when none is left (join_left_behind); a process that never loses a thread never sees them.
#   * an uncaught exception -> 1, sys.exit(n) -> n, mark_exit(n) -> n (a bare `raise SystemExit(n)` at top level passes no hook
#     Python offers -- not sys.excepthook, not the patched sys.exit, and the status is a C local by the time atexit runs: it
#     counts as "nobody said", below, or as "clean" when a finished step marked the exit so.  A failure path that must not be
#     mistaken for success therefore calls mark_exit(n) / sys.exit(n); replacing builtins.SystemExit would break
#     `except SystemExit` around sys.exit);
#   * a step that FINISHED after its exchange was given up (host fall-back of Fast.run / DeviceGroup / step_sharded) marks the
#     exit clean itself (mark_clean_exit), so a program that simply ends after it exits 0; bench.py marks it when its line is out;
#   * none of these (a caller of call_with_deadline that never said its work was done) -> EX_SOFTWARE (70), never a silent 0.
# A recorded code that did not end the process (a SystemExit somebody caught: argparse inside a framework) is forgotten by the next
# deadline call or mark_clean_exit.
_EXIT = {"code": None, "clean": False}
_HOOKS = {"on": False, "excepthook": None, "exit": None}


def _code_of(code):
    return code if isinstance(code, int) else (0 if code is None else 1)


def _install_exit_hooks():
    if _HOOKS["on"]:
        return
    _HOOKS.update(on=True, excepthook=sys.excepthook, exit=sys.exit)
    prev_hook, prev_exit = sys.excepthook, sys.exit

    def excepthook(tp, val, tb):
        _EXIT["code"] = _code_of(val.code) if isinstance(val, SystemExit) else 1
        prev_hook(tp, val, tb)

    def exit_(code=0):
        _EXIT["code"] = _code_of(code)
        prev_exit(code)
    _HOOKS["ours"] = (excepthook, exit_)
    sys.excepthook, sys.exit = excepthook, exit_


def _remove_exit_hooks():
    if not _HOOKS["on"]:
        return
    ours = _HOOKS.get("ours", (None, None))
    if sys.excepthook is ours[0]:          # (somebody who hooked in after us keeps their hook; ours then just chains)
        sys.excepthook = _HOOKS["excepthook"]
    if sys.exit is ours[1]:
        sys.exit = _HOOKS["exit"]
    _HOOKS["on"] = False


def mark_clean_exit():
    """The program's work is done (and reported): a thread left inside RCCL may be abandoned with exit status 0.  Called by the
    library itself when a step finished on the host path after its device exchange was given up, and by bench.py when its line
    is out; an exit code recorded earlier that did not end the process is forgotten."""
    _EXIT["clean"] = True
    _EXIT["code"] = None


def mark_exit(code):
    """The program is about to end with this status although a thread may still sit inside RCCL: record it, so that the exit
    hook leaves with it whatever form the exit then takes (`sys.exit(n)` passes the hook by itself once it is installed; a
    top-level `raise SystemExit(n)` passes NO hook Python offers -- a failure path that ends that way must say so here
    first).  bench.py --require-rccl does, after a run that degraded to the host exchange."""
    _EXIT["code"] = _code_of(code)
    _EXIT["clean"] = False


def exit_status():
    if _EXIT["code"] is not None:
        return _EXIT["code"]
    return 0 if _EXIT["clean"] else 70


def _exit_hook():
    # A thread that is still blocked inside the RCCL / HIP runtime would make interpreter or runtime teardown wait for
    # ever.  Everything the process had to say has been said by now: flush and leave without the teardown, with the status
    # the process was exiting with (never 0 for a run that died or did not say it finished).
    if stuck_threads():
        try:
            sys.stdout.flush()
            sys.stderr.flush()
        finally:
            os._exit(exit_status())


atexit.register(_exit_hook)


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
    rccl_ranks = 0

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


class RcclTransport(HostTransport):
    """The exchange runs inside libfastmc.so on the handle's device buffers (fastmc_comm_gather): ncclAllGather of the
    powers and ncclAllReduce of the histogram over xGMI.  Every exchange runs under a deadline and ends with a collective
    verdict over the rendezvous (did EVERY rank's exchange come back?); on a miss every rank aborts its communicator,
    fetches its own vector and the step -- and every later one -- goes through the host sockets (`degrade`)."""
    name = "rccl"

    def __init__(self, rdzv, rccl_ranks=0):
        super().__init__(rdzv)
        self.rccl_ranks = rccl_ranks      # world size the communicator itself reports
        self.why = ""

    def degrade(self, handle, why):
        self.name, self.why, self.rccl_ranks = "host", f"RCCL exchange given up: {why}", 0
        if self.rank == 0:
            logger.warning(f"RCCL exchange gave no result on every rank ({why}); communicators aborted, host exchange from now on")
        try:
            handle.comm_abort()            # ncclCommAbort: never waits for peers; wakes a blocked exchange
        except Exception as e:
            logger.warning(f"ncclCommAbort: {e}")
        # the exchange that missed its deadline may still be inside the library on this handle: it must be out before the
        # caller touches the handle again (the library's per-handle lock enforces it; this bounds the wait and says why)
        if not join_left_behind(post_abort_timeout()):
            raise ExchangeStuck(f"the aborted RCCL exchange did not return within {post_abort_timeout():g} s ({why})")

    def device_exchange(self, handle, nval, hist_range=None, powers=True):
        """fastmc_comm_gather under the deadline + the collective verdict.  Returns (ok, all_values | None, hist | None);
        after ok == False the transport is a host transport and the caller exchanges the vector `handle.wait()` returns."""
        ok, val = call_with_deadline(lambda: handle.comm_gather(nval, self.world, hist_range, powers=powers), exchange_timeout())
        flags = self.rdzv.all_gather_array(np.array([1 if ok else 0], dtype=np.int32)).ravel()
        if flags.all():
            return True, val[0], val[1]
        bad = np.flatnonzero(flags == 0).tolist()
        self.degrade(handle, val if not ok else f"rank(s) {bad} reported a failed exchange")
        return False, None, None

    def gather(self, local, handle):
        if self.name != "rccl":
            return HostTransport.gather(self, local)
        local = np.ascontiguousarray(local)
        cplx = np.iscomplexobj(local)
        nval = local.size * (2 if cplx else 1)                   # float64 values resident on the device
        ok, allp, _ = self.device_exchange(handle, nval)
        if not ok:
            return HostTransport.gather(self, local)
        allp = allp.reshape(self.world, nval)
        return [allp[r].view(np.complex128) if cplx else allp[r] for r in range(self.world)]

    def gather_with_hist(self, local, handle, hist_range):
        """Powers and the dB histogram of the handle's last run in one call (ncclAllGather + ncclAllReduce)."""
        nval = np.asarray(local).size * (2 if np.iscomplexobj(local) else 1)
        if self.name == "rccl":
            ok, allp, hist = self.device_exchange(handle, nval, hist_range)
            if ok:
                return allp.reshape(self.world, nval), hist
        parts = HostTransport.gather(self, local)
        return np.stack([np.asarray(p).view(np.float64) for p in parts]), self.reduce_hist(handle.histogram(*hist_range))

    def device_hist(self, handle, hist_range):
        """Global dB histogram of the handle's last run: histogram kernel + ncclAllReduce(sum, uint64) on the device."""
        if self.name == "rccl":
            ok, _, hist = self.device_exchange(handle, 1, hist_range, powers=False)
            if ok:
                return hist
        return self.reduce_hist(handle.histogram(*hist_range))


_TRANSPORT = {}          # device -> transport of this process (the communicator belongs to the device)


def make_transport(handle, rdzv, rccl_timeout=None):
    """The transport for this process's device: RCCL when EVERY rank's communicator initialises, else the host path.
    The decision is collective (two rendezvous exchanges); FASTMC_DISABLE_RCCL=1 forces the host path."""
    from . import _lib
    key = handle.device
    if key in _TRANSPORT:
        return _TRANSPORT[key]
    timeout = float(os.environ.get("FASTMC_RCCL_TIMEOUT", "90")) if rccl_timeout is None else rccl_timeout
    if os.environ.get("FASTMC_TEST_STALL_GATHER", "0") == "1":      # ("2" stalls a REAL exchange after its collectives are enqueued)
        # fault injection (tests): behave as if the clique were up; fastmc_comm_gather blocks until it is aborted
        tr = RcclTransport(rdzv, rdzv.world)
        _TRANSPORT[key] = tr
        return tr
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
        mine, val = call_with_deadline(lambda: handle.comm_init(msg[1:], rdzv.world, rdzv.rank), timeout)
        flags = rdzv.all_gather_array(np.array([1 if mine else 0], dtype=np.int32)).ravel()
        ok = bool(flags.all())
        if not ok:
            why = val if not mine else f"rank(s) {np.flatnonzero(flags == 0).tolist()} failed"
            # a half-built clique is ABORTED, never destroyed: ncclCommAbort does not wait for peers that may be gone or
            # still initialising, and an init that returns later sees the abort and drops its communicator (fastmc_comm_init)
            try:
                handle.comm_abort()
            except Exception:
                pass
    if ok:
        tr = RcclTransport(rdzv, handle.comm_world()[0])
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


def step_sharded(handle, transport, seed, real_base, n_real_total, logamp_var=0.0, coherent=False, hist_range=None):
    """One sharded run of realisations [real_base, real_base + n_real_total) with the device generator: this rank's
    contiguous piece, then ONE exchange.  On the RCCL transport the step synchronises once -- the kernels are enqueued
    without waiting (fastmc_run_async) and the all-gather / histogram all-reduce follow on the same stream -- under the
    deadline and the collective verdict of RcclTransport.device_exchange; a missed deadline finishes the step, and all
    later ones, on the host sockets.  Returns (full vector, global histogram | None, info dict)."""
    real0, n_local = shard_range(n_real_total, transport.world, transport.rank)
    nval = 2 * n_local * (2 if coherent else 1)
    info = {"exchange": transport.name, "exchange_device_ms": 0.0}
    t0 = time.perf_counter()
    if transport.name == "rccl":
        handle.run_async(seed, real_base + real0, n_local, logamp_var, coherent)
        ok, allp, hist = transport.device_exchange(handle, nval, hist_range)
        if ok:
            allp = allp.reshape(transport.world, nval)
            parts = [allp[r].view(np.complex128) if coherent else allp[r] for r in range(transport.world)]
            info["exchange_device_ms"] = handle.last_exchange_ms()
            info["wall_ms"] = (time.perf_counter() - t0) * 1e3
            return assemble(parts, complex_out=coherent), hist, info
        # the device's own vector is still resident; its stream carries the aborted collectives, so this wait has a deadline too
        ok, local = call_with_deadline(handle.wait, post_abort_timeout())
        if not ok:
            raise ExchangeStuck(f"the device's results could not be fetched after the RCCL exchange was aborted: {local}")
        info["exchange"] = "host"
    else:
        local = handle.run(seed, real_base + real0, n_local, None, logamp_var, coherent)
    t1 = time.perf_counter()
    parts = transport.gather(local, handle)           # host sockets (a degraded RCCL transport dispatches there too)
    hist = None if hist_range is None else transport.reduce_hist(handle.histogram(*hist_range))
    info["exchange_host_ms"] = (time.perf_counter() - t1) * 1e3
    info["wall_ms"] = (time.perf_counter() - t0) * 1e3
    if getattr(transport, "why", ""):
        mark_clean_exit()          # the step finished on the host path after the device exchange was given up (see the exit hooks)
    return assemble(parts, complex_out=coherent), hist, info


def steps_pipelined(handle, transport, seed, steps, logamp_var=0.0, coherent=False, hist_range=None):
    """Generator: `step_sharded` for each (real_base, n_real_total) of `steps`, in order, with TWO steps in flight on the
    device -- step i + 1's kernels and exchange are enqueued before step i's results are waited for (fastmc_run_queued /
    fastmc_comm_gather_queued / fastmc_queue_wait), so the host side of a step costs the device nothing.  Yields
    (full vector, global histogram | None, info).  Deadline, collective verdict and host fall-back as in `step_sharded`;
    a step whose device exchange failed on any rank is redone by every rank on the host path (same realisations, same numbers)."""
    steps = [(int(b), int(n)) for b, n in steps]
    nbins = hist_range[2] if hist_range is not None else 0
    world, rank = transport.world, transport.rank

    def enqueue(i):
        base, n_total = steps[i]
        real0, n_local = shard_range(n_total, world, rank)
        use_rccl = transport.name == "rccl"
        handle.run_queued(seed, base + real0, n_local, logamp_var, coherent, i & 1, fetch=not use_rccl)
        if use_rccl:
            handle.comm_gather_queued(2 * n_local * (2 if coherent else 1), hist_range, True, i & 1)
        elif hist_range is not None:
            handle.histogram_queued(*hist_range, slot=i & 1)
        return use_rccl

    def drain():
        for slot in (0, 1):
            try:
                handle.queue_wait(slot)
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
            base, n_total = steps[i]
            real0, n_local = shard_range(n_total, world, rank)
            nval = 2 * n_local * (2 if coherent else 1)
            info = {"exchange": "rccl" if mode[i] else "host", "exchange_device_ms": 0.0}
            t0 = time.perf_counter()
            if mode[i]:
                ok, val = call_with_deadline(lambda: handle.queue_wait(i & 1, nval * world, nbins), exchange_timeout())
                flags = transport.rdzv.all_gather_array(np.array([1 if ok else 0], dtype=np.int32)).ravel()
                if not flags.all():
                    bad = np.flatnonzero(flags == 0).tolist()
                    transport.degrade(handle, val if not ok else f"rank(s) {bad} reported a failed exchange")
                    ok2, why2 = call_with_deadline(drain, post_abort_timeout())
                    if not ok2:
                        raise ExchangeStuck(f"queued steps did not drain after the RCCL exchange was aborted: {why2}")
                    mode = {i: enqueue(i)}
                    continue
                allp, hist = val
                allp = allp.reshape(world, nval)
                parts = [allp[r].view(np.complex128) if coherent else allp[r] for r in range(world)]
                info["exchange_device_ms"] = handle.last_exchange_ms()
            else:
                local, lh = handle.queue_wait(i & 1, nval, nbins)
                t1 = time.perf_counter()
                parts = transport.gather(local.view(np.complex128) if coherent else local, handle)
                hist = None if hist_range is None else transport.reduce_hist(lh)
                info["exchange_host_ms"] = (time.perf_counter() - t1) * 1e3
            info["wall_ms"] = (time.perf_counter() - t0) * 1e3
            mode.pop(i, None)
            if getattr(transport, "why", ""):
                mark_clean_exit()          # finished on the host path after the device exchange was given up
            yield assemble(parts, complex_out=coherent), hist, info
            i += 1
        done = True
    finally:
        # a consumer that stops early leaves step i + 1 enqueued (slot busy, on the RCCL path a collective on the stream): empty
        # both slots under the post-abort deadline, so that the next pass over this handle does not find "slot still in flight"
        if not done and mode:
            ok3, why3 = call_with_deadline(drain, post_abort_timeout())
            if not ok3:
                logger.warning(f"queued steps did not drain when the pipelined pass was abandoned: {why3}")


def histogram_sharded(local_hist, transport):
    return transport.reduce_hist(local_hist)
