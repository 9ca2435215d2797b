"""ctypes binding of libfastmc.so (include/fastmc.h).  No torch, no fallback: if the library
is missing or no gfx950 device is visible, the calls raise."""
import ctypes as C
import logging
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FASTMC_LIB selects another build of the same ABI (A/B timing of kernel variants: tools/ab.sh)
LIB_PATH = os.environ.get("FASTMC_LIB") or os.path.join(_HERE, "libfastmc.so")

F64, F32 = 0, 1
KERNEL_PATHS = {"direct": 0, "wave": 1, "chirpz": 2, "lanes50": 3}
AO_MODES = {"NOAO": 0, "AO": 1, "TT": 2, "LGSAO": 3}
PS_NSCALARS = 6

EXPORTS = [
    "fastmc_version", "fastmc_last_error", "fastmc_device_count", "fastmc_create", "fastmc_destroy",
    "fastmc_set_spectrum", "fastmc_set_pupil", "fastmc_set_subharm", "fastmc_run", "fastmc_run_coeffs",
    "fastmc_screens_coeffs", "fastmc_screens", "fastmc_rng_coeffs", "fastmc_rng_logamp", "fastmc_histogram",
    "fastmc_result_stats", "fastmc_last_timing", "fastmc_kernel_path", "fastmc_set_batch", "fastmc_get_batch", "fastmc_powerspec",
    "fastmc_set_layer_screens", "fastmc_temporal_chunk", "fastmc_link_metrics", "fastmc_set_results",
    "fastmc_powerspec_terms", "fastmc_powerspec_set", "fastmc_powerspec_get",
    "fastmc_comm_unique_id", "fastmc_comm_init", "fastmc_comm_init_all", "fastmc_comm_world", "fastmc_comm_gather",
    "fastmc_comm_gather_all", "fastmc_comm_destroy", "fastmc_comm_abort", "fastmc_last_exchange_ms",
    "fastmc_run_async", "fastmc_wait", "fastmc_set_rng_precision", "fastmc_temporal_phases", "fastmc_last_kernels",
    "fastmc_precision", "fastmc_last_result_shape", "fastmc_last_clock", "fastmc_run_queued", "fastmc_comm_gather_queued",
    "fastmc_comm_gather_all_queued", "fastmc_histogram_queued", "fastmc_queue_wait",
    "fastmc_npstream_set_tables", "fastmc_npstream_normals", "fastmc_npstream_logamp", "fastmc_run_npstream",
]


class FastMCError(RuntimeError):
    pass


class LinkQuery(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p0", C.c_double), ("p1", C.c_double)]


LM_FADE, LM_BER_OOK, LM_SEP_QAM = 0, 1, 2


class PsParams(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("n_layers", C.c_int32), ("dx", C.c_double), ("wvl", C.c_double),
        ("L0", C.c_double), ("l0", C.c_double), ("ao_mode", C.c_int32), ("alias", C.c_int32),
        ("noise", C.c_double), ("d_wfs", C.c_double), ("t_loop", C.c_double), ("t_exp", C.c_double),
        ("dtheta", C.c_double * 2), ("cn2", C.c_void_p), ("h", C.c_void_p), ("wind", C.c_void_p),
        ("mask_mode", C.c_int32), ("zmax", C.c_int32), ("modal_mult", C.c_double), ("D_ground", C.c_double),
        ("lf_mask", C.c_void_p), ("pupil_filter", C.c_void_p), ("lgs_z", C.c_void_p), ("simpson_w", C.c_void_p),
        ("pupil_filter_token", C.c_int64),
    ]


_lib = None


def lib():
    """Load libfastmc.so once.  Raises FastMCError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FastMCError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C fast_amd/csrc`).  fast_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    dp, vp, i64, u64 = C.POINTER(C.c_double), C.c_void_p, C.c_int64, C.c_uint64
    L.fastmc_version.restype = C.c_int
    L.fastmc_last_error.restype = C.c_char_p
    L.fastmc_device_count.argtypes = [C.POINTER(C.c_int)]
    L.fastmc_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    L.fastmc_destroy.argtypes = [vp]
    L.fastmc_destroy.restype = None
    L.fastmc_set_spectrum.argtypes = [vp, dp, C.c_double]
    L.fastmc_set_pupil.argtypes = [vp, dp, C.c_int, C.c_double]
    L.fastmc_set_subharm.argtypes = [vp, dp, dp, dp, dp]
    L.fastmc_run.argtypes = [vp, u64, i64, i64, dp, C.c_double, C.c_int, dp]
    L.fastmc_run_coeffs.argtypes = [vp, dp, dp, i64, dp, dp, dp, C.c_int, dp]
    L.fastmc_screens_coeffs.argtypes = [vp, dp, dp, i64, dp, dp, dp]
    L.fastmc_screens.argtypes = [vp, u64, i64, i64, dp]
    L.fastmc_rng_coeffs.argtypes = [vp, u64, i64, dp]
    L.fastmc_rng_logamp.argtypes = [vp, u64, i64, i64, dp]
    L.fastmc_set_layer_screens.argtypes = [vp, dp, C.c_int]
    L.fastmc_temporal_chunk.argtypes = [vp, dp, dp, C.POINTER(C.c_int32), C.c_int, dp, C.c_int, dp]
    L.fastmc_temporal_phases.argtypes = [vp, dp, dp, C.POINTER(C.c_int32), C.c_int, dp]
    L.fastmc_set_results.argtypes = [vp, dp, i64, C.c_int]
    L.fastmc_histogram.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.POINTER(i64)]
    L.fastmc_result_stats.argtypes = [vp, dp, C.c_int, dp]
    L.fastmc_link_metrics.argtypes = [vp, C.c_int, dp, i64, C.POINTER(LinkQuery), C.c_int, dp]
    L.fastmc_last_timing.argtypes = [vp, dp, C.POINTER(i64)]
    L.fastmc_kernel_path.argtypes = [vp, C.c_int]
    L.fastmc_set_batch.argtypes = [vp, C.c_int]
    L.fastmc_get_batch.argtypes = [vp, C.POINTER(C.c_int)]
    L.fastmc_last_kernels.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
    L.fastmc_precision.argtypes = [vp]
    L.fastmc_last_result_shape.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_int)]
    L.fastmc_last_clock.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.fastmc_run_queued.argtypes = [vp, u64, i64, i64, C.c_double, C.c_int, C.c_int, C.c_int]
    L.fastmc_comm_gather_queued.argtypes = [vp, i64, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
    L.fastmc_comm_gather_all_queued.argtypes = [C.POINTER(vp), C.c_int, i64, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
    L.fastmc_histogram_queued.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_int]
    L.fastmc_queue_wait.argtypes = [vp, C.c_int, dp, i64, C.POINTER(i64), C.c_int]
    u64p = C.POINTER(C.c_uint64)
    L.fastmc_npstream_set_tables.argtypes = [C.c_int, dp, u64p, dp]
    L.fastmc_npstream_normals.argtypes = [vp, u64p, i64, dp, u64p, u64p, C.POINTER(C.c_uint32)]
    L.fastmc_npstream_logamp.argtypes = [vp, u64p, i64, C.c_double, dp, u64p, C.POINTER(C.c_uint32)]
    L.fastmc_run_npstream.argtypes = [vp, u64p, i64, i64, i64, C.c_int, dp, u64p, C.POINTER(i64)]
    L.fastmc_set_rng_precision.argtypes = [vp, C.c_int]
    L.fastmc_powerspec.argtypes = [C.c_int, C.POINTER(PsParams), dp, dp, dp, dp, dp, dp]
    L.fastmc_powerspec_terms.argtypes = [C.c_int, C.POINTER(PsParams), dp, dp, dp, dp]
    L.fastmc_powerspec_set.argtypes = [vp, C.POINTER(PsParams), C.c_double, dp, dp]
    L.fastmc_powerspec_get.argtypes = [vp, C.c_int, dp]
    L.fastmc_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.fastmc_comm_init.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int, C.c_int]
    L.fastmc_comm_gather.argtypes = [vp, i64, dp, C.POINTER(i64), C.c_double, C.c_double, C.c_int]
    L.fastmc_comm_init_all.argtypes = [C.POINTER(vp), C.c_int]
    L.fastmc_comm_world.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.fastmc_comm_gather_all.argtypes = [C.POINTER(vp), C.c_int, i64, dp, C.POINTER(i64), C.c_double, C.c_double, C.c_int]
    L.fastmc_comm_destroy.argtypes = [vp]
    L.fastmc_comm_abort.argtypes = [vp]
    L.fastmc_last_exchange_ms.argtypes = [vp, dp]
    L.fastmc_run_async.argtypes = [vp, u64, i64, i64, C.c_double, C.c_int]
    L.fastmc_wait.argtypes = [vp, dp]
    for name in EXPORTS:
        if name not in ("fastmc_last_error", "fastmc_destroy"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def last_error():
    """The message of the calling thread's last failed library call (fastmc_last_error)."""
    return lib().fastmc_last_error().decode()


def _chk(rc):
    if rc < 0:
        raise FastMCError(f"libfastmc error {rc}: {lib().fastmc_last_error().decode()}")
    return rc


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int(0)
    rc = lib().fastmc_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def default_device():
    """LOCAL_RANK (one process per GPU under a launcher); 0 otherwise.  When the launcher ALSO restricted the visible
    devices per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set, e.g. one device each) the local rank is folded onto
    what is visible; otherwise a local rank beyond the visible devices is an error -- several ranks must never land on
    one GPU silently and be reported as N GPUs."""
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if lr > 0:
        n = device_count()
        if n > 0 and lr >= n:
            if os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"):
                lr %= n
            elif os.environ.get("FASTMC_ALLOW_SHARED_DEVICE", "0") not in ("", "0"):
                lr %= n            # tests: several ranks on a 1-GPU box, said so explicitly
            else:
                raise FastMCError(f"LOCAL_RANK={lr} but only {n} GPU(s) visible: start one rank per GPU, restrict the devices "
                                  "per rank with HIP_VISIBLE_DEVICES, or name the device with GPU_DEVICE")
    return lr


_PROMOTED = set()
_NPS_DEVICES = set()      # devices whose library copy of numpy's ziggurat tables has been uploaded


class Handle:
    """One GPU + one (N, Np) Monte-Carlo problem (fastmc_t)."""

    def __init__(self, N, Np, precision="f64", device=None):
        self._h = C.c_void_p()
        self.N, self.Np = int(N), int(Np)
        self.precision = precision
        prec = {"f64": F64, "f32": F32}[precision]
        dev = default_device() if device is None else int(device)
        _chk(lib().fastmc_create(C.byref(self._h), dev, self.N, self.Np, prec))
        self.device = dev
        # what the handle COMPUTES in: float32 exists on the wave family's fixed grids and the direct family; elsewhere a
        # float32 request runs the float64 kernels (fastmc_create) -- the label follows the arithmetic, and says so once
        self.precision_requested = precision
        self.precision = self.effective_precision()
        if self.precision != precision and (self.N, precision) not in _PROMOTED:
            _PROMOTED.add((self.N, precision))
            logging.getLogger(__name__).warning(f"GPU_PRECISION '{precision}' has no kernels on a {self.N}^2 grid: computing in {self.precision}")

    def close(self):
        if self._h:
            lib().fastmc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_spectrum(self, powerspec, df):
        ps = _f64(powerspec)
        assert ps.shape == (self.N, self.N)
        _chk(lib().fastmc_set_spectrum(self._h, _dptr(ps), float(df)))

    def set_pupil(self, W, crop_lo, dx):
        W = _f64(W)
        assert W.shape == (self.Np, self.Np)
        _chk(lib().fastmc_set_pupil(self._h, _dptr(W), int(crop_lo), float(dx)))

    def set_subharm(self, powerspec_sh, fx, fy, df):
        if powerspec_sh is None:
            _chk(lib().fastmc_set_subharm(self._h, None, None, None, None))
            return
        a, b, c, d = _f64(powerspec_sh), _f64(fx), _f64(fy), _f64(df)
        assert a.shape == (3, 3, 3) and b.shape == (3, 3, 3) and c.shape == (3, 3, 3) and d.shape == (3,)
        _chk(lib().fastmc_set_subharm(self._h, _dptr(a), _dptr(b), _dptr(c), _dptr(d)))

    def run(self, seed, real0, n_real, logamp=None, logamp_var=0.0, coherent=False):
        la = _f64(logamp)
        if la is not None:
            assert la.shape == (2 * n_real,)
        out = np.empty(2 * n_real * (2 if coherent else 1), dtype=np.float64)
        _chk(lib().fastmc_run(self._h, int(seed) & (2 ** 64 - 1), int(real0), int(n_real), _dptr(la), float(logamp_var),
                              int(bool(coherent)), _dptr(out)))
        return out.view(np.complex128) if coherent else out

    def run_async(self, seed, real0, n_real, logamp_var=0.0, coherent=False):
        """Enqueue the run and return at once: the results stay on the device (log-amplitudes drawn there).  `wait()`
        fetches them; an exchange (`comm_gather`, `_lib.comm_gather_all`) is ordered behind the kernels on the same
        stream, so a sharded step synchronises once."""
        _chk(lib().fastmc_run_async(self._h, int(seed) & (2 ** 64 - 1), int(real0), int(n_real), float(logamp_var), int(bool(coherent))))

    def wait(self, fetch=True):
        """Wait for the handle's stream; with fetch, return the results of the last run as `run` would have."""
        if not fetch:
            _chk(lib().fastmc_wait(self._h, None))
            return None
        # the library says what is resident (after run, run_async or set_results alike): never a stale shape of our own
        n_it, coh = C.c_int64(0), C.c_int(0)
        _chk(lib().fastmc_last_result_shape(self._h, C.byref(n_it), C.byref(coh)))
        if n_it.value <= 0:
            raise FastMCError("wait(fetch=True): no results are resident on this handle")
        coherent = bool(coh.value)
        out = np.empty(n_it.value * (2 if coherent else 1), dtype=np.float64)
        _chk(lib().fastmc_wait(self._h, _dptr(out)))
        return out.view(np.complex128) if coherent else out

    # ---- two steps in flight (fastmc.h: fastmc_run_queued ...)
    def run_queued(self, seed, real0, n_real, logamp_var=0.0, coherent=False, slot=0, fetch=True):
        """Enqueue a step on one of the handle's two slots and return; with fetch, its own result vector lands on the slot."""
        _chk(lib().fastmc_run_queued(self._h, int(seed) & (2 ** 64 - 1), int(real0), int(n_real), float(logamp_var), int(bool(coherent)),
                                     int(slot), int(bool(fetch))))

    def comm_gather_queued(self, n_local, hist_range=None, powers=True, slot=0):
        lo, hi, nb = hist_range if hist_range is not None else (0.0, 1.0, 0)
        _chk(lib().fastmc_comm_gather_queued(self._h, int(n_local), int(bool(powers)), float(lo), float(hi), int(nb), int(slot)))

    def histogram_queued(self, lo, hi, nbins, slot=0):
        _chk(lib().fastmc_histogram_queued(self._h, float(lo), float(hi), int(nbins), int(slot)))

    def queue_wait(self, slot, n_out=0, hist_bins=0):
        """Wait for the slot's step: (values | None, histogram | None).  n_out: capacity of the value buffer in float64."""
        out = np.empty(int(n_out), dtype=np.float64) if n_out else None
        hist = np.zeros(int(hist_bins) + 2, dtype=np.int64) if hist_bins else None
        n = _chk(lib().fastmc_queue_wait(self._h, int(slot), _dptr(out), int(n_out), None if hist is None else hist.ctypes.data_as(C.POINTER(C.c_int64)),
                                         0 if hist is None else hist.size))
        return (None if out is None else out[:n]), hist

    # ---- numpy's normal stream on the device (fastmc.h: fastmc_npstream_*; fast_amd/npnormal.py)
    def _npstream_ready(self):
        if self.device not in _NPS_DEVICES:
            from . import npnormal
            wi, ki, fi = npnormal.get_tables()
            wi, fi, ki = np.ascontiguousarray(wi, dtype=np.float64), np.ascontiguousarray(fi, dtype=np.float64), np.ascontiguousarray(ki, dtype=np.uint64)
            _chk(lib().fastmc_npstream_set_tables(self.device, _dptr(wi), ki.ctypes.data_as(C.POINTER(C.c_uint64)), _dptr(fi)))
            _NPS_DEVICES.add(self.device)

    @staticmethod
    def _u64p(a):
        return a.ctypes.data_as(C.POINTER(C.c_uint64))

    def npstream_normals(self, state_words, n, fetch=True):
        """`Generator(PCG64 at state_words).normal(size=n)` drawn on the device -> (values | None, state after (lo, hi), words
        consumed, overflow flags)."""
        self._npstream_ready()
        sw = np.ascontiguousarray(state_words, dtype=np.uint64)
        out = np.empty(int(n)) if fetch else None
        after, cons, ovf = np.zeros(2, dtype=np.uint64), C.c_uint64(0), C.c_uint32(0)
        _chk(lib().fastmc_npstream_normals(self._h, self._u64p(sw), int(n), _dptr(out), self._u64p(after), C.byref(cons), C.byref(ovf)))
        return out, after, cons.value, ovf.value

    def npstream_logamp(self, state_words, n_iter, logamp_var):
        self._npstream_ready()
        sw = np.ascontiguousarray(state_words, dtype=np.uint64)
        la = np.empty(int(n_iter))
        after, ovf = np.zeros(2, dtype=np.uint64), C.c_uint32(0)
        _chk(lib().fastmc_npstream_logamp(self._h, self._u64p(sw), int(n_iter), float(logamp_var), _dptr(la), self._u64p(after), C.byref(ovf)))
        return la, after, ovf.value

    def run_npstream(self, state_words, n_chunks, chunk_real, logamp_offset, coherent=False):
        """Chunks of the Monte-Carlo loop with numpy-stream coefficients -> (I[n_chunks][2 chunk_real], state after, bad chunk | -1)."""
        self._npstream_ready()
        sw = np.ascontiguousarray(state_words, dtype=np.uint64)
        out = np.empty((int(n_chunks), 2 * int(chunk_real) * (2 if coherent else 1)))
        after, bad = np.zeros(2, dtype=np.uint64), C.c_int64(-1)
        _chk(lib().fastmc_run_npstream(self._h, self._u64p(sw), int(n_chunks), int(chunk_real), int(logamp_offset), int(bool(coherent)),
                                       _dptr(out), self._u64p(after), C.byref(bad)))
        return (out.view(np.complex128) if coherent else out), after, bad.value

    def run_coeffs(self, coeff_re, coeff_im, logamp, coherent=False, sh_re=None, sh_im=None):
        cr, ci, la = _f64(coeff_re), _f64(coeff_im), _f64(logamp)
        n_real = cr.shape[0]
        assert cr.shape == (n_real, self.N, self.N) and ci.shape == cr.shape and la.shape == (2 * n_real,)
        sr, si = _f64(sh_re), _f64(sh_im)
        out = np.empty(2 * n_real * (2 if coherent else 1), dtype=np.float64)
        _chk(lib().fastmc_run_coeffs(self._h, _dptr(cr), _dptr(ci), n_real, _dptr(sr), _dptr(si), _dptr(la),
                                     int(bool(coherent)), _dptr(out)))
        return out.view(np.complex128) if coherent else out

    def screens_coeffs(self, coeff_re, coeff_im, sh_re=None, sh_im=None):
        cr, ci = _f64(coeff_re), _f64(coeff_im)
        n_real = cr.shape[0]
        sr, si = _f64(sh_re), _f64(sh_im)
        phs = np.empty((2 * n_real, self.Np, self.Np), dtype=np.float64)
        _chk(lib().fastmc_screens_coeffs(self._h, _dptr(cr), _dptr(ci), n_real, _dptr(sr), _dptr(si), _dptr(phs)))
        return phs

    def screens(self, seed, real0, n_real):
        phs = np.empty((2 * n_real, self.Np, self.Np), dtype=np.float64)
        _chk(lib().fastmc_screens(self._h, int(seed) & (2 ** 64 - 1), int(real0), int(n_real), _dptr(phs)))
        return phs

    def rng_coeffs(self, seed, real):
        out = np.empty((self.N, self.N), dtype=np.complex128)
        _chk(lib().fastmc_rng_coeffs(self._h, int(seed) & (2 ** 64 - 1), int(real), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def rng_logamp(self, seed, iter0, n_iter):
        out = np.empty(n_iter, dtype=np.float64)
        _chk(lib().fastmc_rng_logamp(self._h, int(seed) & (2 ** 64 - 1), int(iter0), int(n_iter), _dptr(out)))
        return out

    def set_layer_screens(self, screens):
        sc = _f64(screens)
        assert sc.ndim == 3 and sc.shape[1:] == (self.N, self.N)
        _chk(lib().fastmc_set_layer_screens(self._h, _dptr(sc), sc.shape[0]))
        self._n_layers = sc.shape[0]

    def temporal_chunk(self, xs, ys, roll, logamp, coherent=False):
        xs, ys, la = _f64(xs), _f64(ys), _f64(logamp)
        roll = np.ascontiguousarray(roll, dtype=np.int32)
        L, M, Np = xs.shape
        assert ys.shape == xs.shape and Np == self.Np and roll.shape == (L, 2, M) and la.shape == (M,)
        out = np.empty(M * (2 if coherent else 1))
        _chk(lib().fastmc_temporal_chunk(self._h, _dptr(xs), _dptr(ys), roll.ctypes.data_as(C.POINTER(C.c_int32)), M,
                                         _dptr(la), int(bool(coherent)), _dptr(out)))
        return out.view(np.complex128) if coherent else out

    def temporal_phases(self, xs, ys, roll):
        """(M, Np, Np) phases of one frozen-flow chunk (the reference's Fast.phs after compute_phs_temporal)."""
        xs, ys = _f64(xs), _f64(ys)
        roll = np.ascontiguousarray(roll, dtype=np.int32)
        L, M, Np = xs.shape
        assert ys.shape == xs.shape and Np == self.Np and roll.shape == (L, 2, M)
        phs = np.empty((M, Np, Np))
        _chk(lib().fastmc_temporal_phases(self._h, _dptr(xs), _dptr(ys), roll.ctypes.data_as(C.POINTER(C.c_int32)), M, _dptr(phs)))
        return phs

    def histogram(self, lo_db, hi_db, nbins):
        bins = np.zeros(nbins + 2, dtype=np.int64)
        _chk(lib().fastmc_histogram(self._h, float(lo_db), float(hi_db), int(nbins), bins.ctypes.data_as(C.POINTER(C.c_int64))))
        return bins

    def set_results(self, values):
        """Make `values` (float64 powers or complex128 amplitudes) the resident results of the handle."""
        v = np.ascontiguousarray(values)
        coherent = np.iscomplexobj(v)
        v = v.astype(np.complex128 if coherent else np.float64, copy=False)
        _chk(lib().fastmc_set_results(self._h, v.view(np.float64).ctypes.data_as(C.POINTER(C.c_double)), v.size, int(coherent)))

    def result_stats(self, thresholds=()):
        """Device-side statistics of the last run: dict(n, mean, scintillation_index, mean_dB_rel,
        avg_dB_rel, min, max, fade_prob[...]) -- FastResult's summaries without the vector."""
        thr = _f64(np.asarray(thresholds, dtype=float).ravel())
        st = np.zeros(6 + len(thr))
        _chk(lib().fastmc_result_stats(self._h, _dptr(thr) if len(thr) else None, len(thr), _dptr(st)))
        n, s1, s2 = st[0], st[1], st[2]
        mean = s1 / n
        return {"n": int(n), "mean": mean, "scintillation_index": (s2 / n) / mean ** 2 - 1.0, "mean_dB_rel": st[3] / n,
                "avg_dB_rel": 10 * np.log10(mean), "min": st[4], "max": st[5], "fade_prob": st[6:] / n}

    def last_timing(self):
        t = np.zeros(4)
        n = np.zeros(4, dtype=np.int64)
        _chk(lib().fastmc_last_timing(self._h, _dptr(t), n.ctypes.data_as(C.POINTER(C.c_int64))))
        return {"total_ms": t[0], "rows_ms": t[1], "cols_ms": t[2], "finalize_ms": t[3],
                "rows_launches": int(n[1]), "cols_launches": int(n[2]), "finalize_launches": int(n[3])}

    def powerspec_get(self, which):
        """(N, N) grid left on the device by powerspec_set: 'powerspec', 'logamp_powerspec' or 'lf_mask'."""
        out = np.empty((self.N, self.N))
        _chk(lib().fastmc_powerspec_get(self._h, {"powerspec": 0, "logamp_powerspec": 1, "lf_mask": 2}[which], _dptr(out)))
        return out

    def kernel_path(self, force=-1):
        return _chk(lib().fastmc_kernel_path(self._h, int(force)))

    def last_kernels(self):
        """(rows, cols): names of the row / column kernels the handle launched last, as c++filt prints them."""
        r, c = C.create_string_buffer(128), C.create_string_buffer(128)
        _chk(lib().fastmc_last_kernels(self._h, r, c, 128))
        return r.value.decode(), c.value.decode()

    def effective_precision(self):
        """'f64' or 'f32': what the handle computes in (a float32 request on a grid without float32 kernels runs float64)."""
        return {F64: "f64", F32: "f32"}[_chk(lib().fastmc_precision(self._h))]

    def set_batch(self, batch):
        _chk(lib().fastmc_set_batch(self._h, int(batch)))

    def get_batch(self):
        """Realisations per launch of this handle's runs as it stands (fastmc_get_batch): a run of n is n // batch launches of
        `batch` and one of the remainder."""
        b = C.c_int(0)
        _chk(lib().fastmc_get_batch(self._h, C.byref(b)))
        return int(b.value)

    def last_clock(self):
        """(GHz, span in microseconds) of the shader clock inside the last row-kernel launch (fastmc_last_clock), or None when no
        launch of this handle has stamped it (only the wave family's row kernel does)."""
        g, us = C.c_double(0.0), C.c_double(0.0)
        rc = lib().fastmc_last_clock(self._h, C.byref(g), C.byref(us))
        if rc == -4:           # FASTMC_ESTATE: nothing stamped yet
            return None
        _chk(rc)
        return g.value, us.value

    def set_rng_precision(self, precision):
        """Device generator: 'f64' (the reference's 53-bit normals and float64 colouring: what fastmc_create leaves on a float64
        handle; fused into the row kernels of every FFT family where its 6 KB of tables fit the LDS, staged in device memory
        otherwise) or 'f32' (the opt-in float32 draw; what a float32 handle starts with)."""
        _chk(lib().fastmc_set_rng_precision(self._h, {"f64": F64, "f32": F32}[precision]))

    # ---- RCCL (the communicator belongs to the handle's DEVICE and outlives the handle)
    def comm_init(self, unique_id, world_size, rank):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _chk(lib().fastmc_comm_init(self._h, buf, int(world_size), int(rank)))

    def comm_world(self):
        """(world size, rank) of the communicator of this handle's device; (0, -1) when there is none."""
        w, r = C.c_int(0), C.c_int(-1)
        _chk(lib().fastmc_comm_world(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def comm_destroy(self):
        _chk(lib().fastmc_comm_destroy(self._h))

    def comm_abort(self):
        """ncclCommAbort on this device's communicator (never waits for peers); wakes a blocked exchange of the device."""
        _chk(lib().fastmc_comm_abort(self._h))

    def last_exchange_ms(self):
        ms = C.c_double(0.0)
        _chk(lib().fastmc_last_exchange_ms(self._h, C.byref(ms)))
        return ms.value

    def comm_gather(self, n_local, world_size, hist_range=None, powers=True):
        allp = np.empty(n_local * world_size, dtype=np.float64) if powers else None
        hist = None
        lo = hi = 0.0
        nb = 1
        if hist_range is not None:
            lo, hi, nb = hist_range
            hist = np.zeros(nb + 2, dtype=np.int64)
        _chk(lib().fastmc_comm_gather(self._h, int(n_local), _dptr(allp),
                                      None if hist is None else hist.ctypes.data_as(C.POINTER(C.c_int64)),
                                      float(lo), float(hi), int(nb)))
        return allp, hist


def centred_fft2(g, device=None, inverse=False):
    """Centred 2-D DFT of a square complex array on the GPU: numpy's fftshift(fft2(fftshift(g))), or with
    inverse=True ifftshift(ifft2(ifftshift(g))) -- the kernels of the Monte-Carlo path with the window
    set to the whole grid, a unit spectrum and host-supplied 'coefficients' g.  Used by the analytic
    mean-irradiance path (fast/fast.py:736-761: aotools ft2 / ift2)."""
    g = np.asarray(g)
    N = g.shape[0]
    assert g.shape == (N, N)
    h = Handle(N, N, "f64", device)
    try:
        h.set_pupil(np.ones((N, N)), 0, 1.0)
        h.set_spectrum(np.ones((N, N)), 1.0)
        x = np.conj(g) if inverse else g
        phs = h.screens_coeffs(np.ascontiguousarray(x.real, dtype=float)[None], np.ascontiguousarray(x.imag, dtype=float)[None])
    finally:
        h.close()
    out = phs[0] + 1j * phs[1]
    if not inverse:
        return out
    out = np.conj(out) / N ** 2                   # = fftshift(ifft2(fftshift(g)))
    if N % 2:
        # odd N: ifftshift = one more sample of roll than fftshift, on the way in (a linear phase
        # after the transform) and on the way out
        n = np.fft.fftshift(np.arange(N))
        ph = np.exp(2j * np.pi * n / N)
        out = np.roll(out * ph[:, None] * ph[None, :], (1, 1), axis=(0, 1))
    return out


def link_metrics(queries, samples=None, handle=None, device=0):
    """fastmc_link_metrics: queries = [(kind, p0, p1), ...] over `samples` (uploaded) or, when samples
    is None, over the last run's results resident on `handle`'s device.  Returns (n_queries, 4)."""
    q = (LinkQuery * len(queries))(*[LinkQuery(int(k), float(a), float(b)) for k, a, b in queries])
    out = np.zeros((len(queries), 4))
    if samples is not None:
        x = _f64(np.asarray(samples, dtype=float).ravel())
        _chk(lib().fastmc_link_metrics(None, int(device), _dptr(x), len(x), q, len(queries), _dptr(out)))
    else:
        if handle is None:
            raise FastMCError("link_metrics needs samples or a handle")
        _chk(lib().fastmc_link_metrics(handle._h, 0, None, 0, q, len(queries), _dptr(out)))
    return out


def _handle_array(handles):
    return (C.c_void_p * len(handles))(*[h._h for h in handles])


def comm_init_all(handles):
    """ncclCommInitAll over the devices of `handles` (one handle per device): handles[i] becomes rank i."""
    _chk(lib().fastmc_comm_init_all(_handle_array(handles), len(handles)))


def comm_gather_all(handles, n_local, hist_range=None, powers=True):
    """Grouped all-gather of every handle's last-run results (n_local float64 each) and all-reduce of the
    dB histogram, issued from this thread for all handles; returns (all_values | None, hist | None)."""
    allp = np.empty(n_local * len(handles), dtype=np.float64) if powers else None
    hist = None
    lo = hi = 0.0
    nb = 1
    if hist_range is not None:
        lo, hi, nb = hist_range
        hist = np.zeros(nb + 2, dtype=np.int64)
    _chk(lib().fastmc_comm_gather_all(_handle_array(handles), len(handles), int(n_local), _dptr(allp),
                                      None if hist is None else hist.ctypes.data_as(C.POINTER(C.c_int64)),
                                      float(lo), float(hi), int(nb)))
    return allp, hist


def comm_gather_all_queued(handles, n_local, hist_range=None, powers=True, slot=0):
    """The exchange of a queued step for all handles of this process (after run_queued on `slot` of each): enqueue and return;
    handles[0].queue_wait(slot, n_local * len(handles), nbins) collects, the other handles' queue_wait only wait."""
    lo, hi, nb = hist_range if hist_range is not None else (0.0, 1.0, 0)
    _chk(lib().fastmc_comm_gather_all_queued(_handle_array(handles), len(handles), int(n_local), int(bool(powers)), float(lo), float(hi),
                                             int(nb), int(slot)))


def comm_unique_id():
    buf = (C.c_uint8 * 128)()
    _chk(lib().fastmc_comm_unique_id(buf))
    return bytes(buf)


def mask_spec(modal, modal_mult, zmax):
    """(mask_mode, zmax, modal_mult) of fastmc_ps_params for the reference's mask_lf arguments."""
    if not modal:
        return 1, 0, 1.0
    if zmax is None:
        return 2, 0, float(modal_mult)
    return 3, int(zmax), float(modal_mult)


def _ps_params(N, dx, wvl, L0, l0, ao_mode, alias, noise, d_wfs, t_loop, t_exp, dtheta, cn2, h, wind, pupil_filter,
               simpson_w, lf_mask, modal, modal_mult, zmax, D_ground, lgs_z, pupil_filter_token=0):
    """fastmc_ps_params for the two power-spectrum entry points; returns (struct, arrays it points into)."""
    cn2, h, wind = _f64(cn2), _f64(h), _f64(wind)
    Lr = len(cn2)
    mask, pf, z, w = _f64(lf_mask), _f64(pupil_filter), _f64(lgs_z), _f64(simpson_w)
    assert w.shape == (N,) and wind.shape == (Lr, 2)
    p = PsParams()
    p.N, p.n_layers, p.dx, p.wvl, p.L0, p.l0 = int(N), Lr, float(dx), float(wvl), float(L0), float(l0)
    p.ao_mode, p.alias, p.noise, p.d_wfs = AO_MODES[ao_mode], int(bool(alias)), float(noise), float(d_wfs)
    p.t_loop, p.t_exp = float(t_loop), float(t_exp)
    p.dtheta[0], p.dtheta[1] = float(dtheta[0]), float(dtheta[1])
    p.cn2, p.h, p.wind = cn2.ctypes.data, h.ctypes.data, wind.ctypes.data
    p.simpson_w = w.ctypes.data
    if mask is None:
        p.mask_mode, p.zmax, p.modal_mult = mask_spec(modal, modal_mult, zmax)
        p.lf_mask = None
    else:
        assert mask.shape == (N, N)
        p.mask_mode, p.zmax, p.modal_mult = 0, 0, 1.0
        p.lf_mask = mask.ctypes.data
    p.D_ground = float(D_ground)
    p.pupil_filter = None if pf is None else pf.ctypes.data
    p.pupil_filter_token = int(pupil_filter_token) if pf is not None else 0
    p.lgs_z = None if z is None else z.ctypes.data
    return p, (cn2, h, wind, mask, pf, z, w)


def powerspec(N, dx, wvl, L0, l0, ao_mode, alias, noise, d_wfs, t_loop, t_exp, dtheta, cn2, h, wind, pupil_filter,
              simpson_w, lf_mask=None, modal=False, modal_mult=1, zmax=None, D_ground=0.0, lgs_z=None,
              per_layer=False, device=None):
    """fastmc_powerspec: mask_lf, AO-residual PSD grid and Simpson scalars on the GPU.
    `lf_mask=None`: the mask is evaluated on the device from (modal, modal_mult, zmax, D_ground);
    otherwise the given (N, N) grid is used.  Returns dict(powerspec, powerspec_per_layer|None,
    logamp_powerspec, lf_mask, scalars..., kernel_ms)."""
    p, keep = _ps_params(N, dx, wvl, L0, l0, ao_mode, alias, noise, d_wfs, t_loop, t_exp, dtheta, cn2, h, wind, pupil_filter,
                         simpson_w, lf_mask, modal, modal_mult, zmax, D_ground, lgs_z)
    Lr = p.n_layers
    ps = np.empty((N, N))
    la = np.empty((N, N))
    mo = np.empty((N, N))
    pl = np.empty((Lr, N, N)) if per_layer else None
    sc = np.empty(PS_NSCALARS + Lr)
    ms = C.c_double(0.0)
    dev = default_device() if device is None else int(device)
    _chk(lib().fastmc_powerspec(dev, C.byref(p), _dptr(ps), _dptr(pl), _dptr(la), _dptr(mo), _dptr(sc), C.byref(ms)))
    return {"powerspec": ps, "powerspec_per_layer": pl, "logamp_powerspec": la, "lf_mask": mo,
            "aniso_servo_error": sc[0], "alias_error": sc[1], "noise_error": sc[2], "fitting_error": sc[3],
            "phs_var": sc[4], "logamp_var": sc[5], "phs_var_weights": sc[6:].copy(), "kernel_ms": ms.value}


def powerspec_set(handle, df, dx, wvl, L0, l0, ao_mode, alias, noise, d_wfs, t_loop, t_exp, dtheta, cn2, h, wind, pupil_filter,
                  simpson_w, lf_mask=None, modal=False, modal_mult=1, zmax=None, D_ground=0.0, lgs_z=None, pupil_filter_token=0):
    """fastmc_powerspec_set: the residual PSD evaluated on the handle's device and left there as its spectrum
    (no N x N grid crosses PCIe); returns the Simpson scalars and the kernel time.  The grids are fetched on demand
    with Handle.powerspec_get."""
    p, keep = _ps_params(handle.N, dx, wvl, L0, l0, ao_mode, alias, noise, d_wfs, t_loop, t_exp, dtheta, cn2, h, wind, pupil_filter,
                         simpson_w, lf_mask, modal, modal_mult, zmax, D_ground, lgs_z, pupil_filter_token)
    sc = np.empty(PS_NSCALARS + p.n_layers)
    ms = C.c_double(0.0)
    _chk(lib().fastmc_powerspec_set(handle._h, C.byref(p), float(df), _dptr(sc), C.byref(ms)))
    return {"aniso_servo_error": sc[0], "alias_error": sc[1], "noise_error": sc[2], "fitting_error": sc[3],
            "phs_var": sc[4], "logamp_var": sc[5], "phs_var_weights": sc[6:].copy(), "kernel_ms": ms.value}


def powerspec_terms(N, dx, wvl, L0, l0, ao_mode, alias, noise, d_wfs, t_loop, t_exp, dtheta, cn2, h, wind, pupil_filter,
                    simpson_w, lf_mask=None, modal=False, modal_mult=1, zmax=None, D_ground=0.0, lgs_z=None, device=None):
    """fastmc_powerspec_terms: the (L, N, N) von Karman, G_AO and alias grids and the (N, N) noise grid."""
    p, keep = _ps_params(N, dx, wvl, L0, l0, ao_mode, alias, noise, d_wfs, t_loop, t_exp, dtheta, cn2, h, wind, pupil_filter,
                         simpson_w, lf_mask, modal, modal_mult, zmax, D_ground, lgs_z)
    Lr = p.n_layers
    turb, g, al, no = np.empty((Lr, N, N)), np.empty((Lr, N, N)), np.empty((Lr, N, N)), np.empty((N, N))
    dev = default_device() if device is None else int(device)
    _chk(lib().fastmc_powerspec_terms(dev, C.byref(p), _dptr(turb), _dptr(g), _dptr(al), _dptr(no)))
    return {"turb_powerspec": turb, "G_ao": g, "alias_powerspec": al, "noise_powerspec": no}
