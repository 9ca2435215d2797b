"""numpy's `Generator.normal` stream for the device generator `GPU_RNG: 'numpy'` (fast_amd/csrc/fmc_npstream.h).

The reference draws from `funcs._R = numpy.random.default_rng(seed)` (fast/funcs.py:21, 352-365).  Reproducing its numbers for a
given SEED means reproducing that stream: PCG64 -> numpy's 256-layer ziggurat.  The algorithm is in fmc_npstream.h; what lives
here is what must come from the numpy that is installed rather than from our source:

  * the ziggurat tables wi / ki (and fi): read out of numpy through a CRAFTED bit generator -- an object with the `.capsule` /
    `.lock` numpy.random.Generator asks for, whose 64-bit outputs we choose.  The word (idx, sign 0, rabs 1) makes
    `standard_normal()` return wi[idx] itself; a bisection on rabs for the first value that draws a second word finds ki[idx];
    fi[idx] = exp(-x_idx^2 / 2) at the layer edge x_idx = wi[idx] 2^52 (fi[0] = 1), which only matters to the last bit of a
    comparison;
  * a self-check: the restated stream (pure Python, `restated_normals`) against `numpy.random.Generator(PCG64).normal` itself,
    before the tables go to a device.  If numpy ever changes its algorithm the check fails and `GPU_RNG: 'numpy'` refuses to run
    (the caller falls back to host draws with a warning) instead of returning different numbers.
No GPU is needed for any of this (tests/test_npnormal.py runs it on the CPU)."""
import ctypes
import math
import threading

import numpy as np

PCG_MULT = (0x2360ED051FC65DA4 << 64) | 0x4385DF649FCCF645
M128 = (1 << 128) - 1
M64 = (1 << 64) - 1
ZIG_R, ZIG_INV_R = 3.6541528853610087963519472518, 0.27366123732975827203338247596


class _BitGen(ctypes.Structure):
    _fields_ = [("state", ctypes.c_void_p), ("next_uint64", ctypes.c_void_p), ("next_uint32", ctypes.c_void_p),
                ("next_double", ctypes.c_void_p), ("next_raw", ctypes.c_void_p)]


class CraftedBitGenerator:
    """A bit generator whose outputs are the words in `feed` (then zeros).  numpy.random.Generator needs `.capsule`, a PyCapsule
    named "BitGenerator" around a bitgen_t, and `.lock`."""

    def __init__(self):
        self.lock = threading.Lock()
        self.feed, self.pos = [], 0
        self._u64 = ctypes.CFUNCTYPE(ctypes.c_uint64, ctypes.c_void_p)(self._next64)
        self._u32 = ctypes.CFUNCTYPE(ctypes.c_uint32, ctypes.c_void_p)(lambda s: self._next64(s) >> 32)
        self._dbl = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p)(lambda s: (self._next64(s) >> 11) * (1.0 / 9007199254740992.0))
        self._bg = _BitGen(None, ctypes.cast(self._u64, ctypes.c_void_p), ctypes.cast(self._u32, ctypes.c_void_p),
                           ctypes.cast(self._dbl, ctypes.c_void_p), ctypes.cast(self._u64, ctypes.c_void_p))
        new = ctypes.pythonapi.PyCapsule_New
        new.restype = ctypes.py_object
        new.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
        self.capsule = new(ctypes.addressof(self._bg), b"BitGenerator", None)

    def _next64(self, _):
        v = self.feed[self.pos] if self.pos < len(self.feed) else 0
        self.pos += 1
        return v

    def load(self, words):
        self.feed, self.pos = list(words), 0


def _word(idx, sign, rabs):
    return (((rabs << 1) | sign) << 8) | idx


_TABLES = None
_LOCK = threading.Lock()


def extract_tables():
    """(wi float64[256], ki uint64[256], fi float64[256]) of the installed numpy's ziggurat (see the module docstring)."""
    bg = CraftedBitGenerator()
    g = np.random.Generator(bg)
    F = M64
    wi = np.zeros(256)
    ki = np.zeros(256, dtype=np.uint64)
    for idx in range(256):
        # rabs = 1; should the word take the slow path (ki[idx] <= 1) the follow-ups make it accept: u = 0 in a wedge, (0, ~1) in the tail
        bg.load([_word(idx, 0, 1), 0, F, 0, F])
        wi[idx] = g.standard_normal()

        def slow(r):
            bg.load([_word(idx, 0, r), 0, F, 0, F, 0, F, 0, 0])
            g.standard_normal()
            return bg.pos > 1
        if slow(0):
            ki[idx] = 0
            continue
        a, b = 0, (1 << 52) - 1
        if not slow(b):
            ki[idx] = 1 << 52
            continue
        while b - a > 1:
            m = (a + b) // 2
            if slow(m):
                b = m
            else:
                a = m
        ki[idx] = b
    fi = np.exp(-0.5 * (wi * 2.0 ** 52) ** 2)
    fi[0] = 1.0
    return wi, ki, fi


def pcg64_words(state, inc):
    """Generator of the 64-bit outputs of numpy's PCG64 from (state, inc) as `bit_generator.state['state']` holds them."""
    while True:
        state = (state * PCG_MULT + inc) & M128
        x = ((state >> 64) ^ state) & M64
        rot = state >> 122
        yield ((x >> rot) | (x << ((64 - rot) & 63))) & M64


def pcg64_advance(state, inc, k):
    """The state k steps on (O(log k))."""
    a, c, A, C = PCG_MULT, inc, 1, 0
    while k:
        if k & 1:
            A, C = (A * a) & M128, (C * a + c) & M128
        c = ((a + 1) * c) & M128
        a = (a * a) & M128
        k >>= 1
    return (A * state + C) & M128


def restated_normals(state, inc, n, tables=None):
    """numpy/random/src/distributions/distributions.c: random_standard_normal, restated (pure Python: tests and the self-check
    only).  Returns (values float64[n], words consumed)."""
    wi, ki, fi = tables or get_tables(check=False)
    g = pcg64_words(state, inc)
    used = [0]

    def nxt():
        used[0] += 1
        return next(g)

    def nd():
        return (nxt() >> 11) * (1.0 / 9007199254740992.0)
    out = np.empty(n)
    i = 0
    while i < n:
        r = nxt()
        idx = r & 0xff
        rabs = (r >> 9) & 0x000fffffffffffff
        x = rabs * wi[idx]
        if (r >> 8) & 1:
            x = -x
        if rabs < ki[idx]:
            out[i] = x
            i += 1
            continue
        if idx == 0:
            while True:
                xx = -ZIG_INV_R * math.log1p(-nd())
                yy = -math.log1p(-nd())
                if yy + yy > xx * xx:
                    out[i] = -(ZIG_R + xx) if ((rabs >> 8) & 1) else ZIG_R + xx
                    i += 1
                    break
        elif (fi[idx - 1] - fi[idx]) * nd() + fi[idx] < math.exp(-0.5 * x * x):
            out[i] = x
            i += 1
    return out, used[0]


def get_tables(check=True):
    """The tables, extracted once per process; with check, the restated stream must equal numpy's own on 30 000 draws."""
    global _TABLES
    with _LOCK:
        if _TABLES is None:
            t = extract_tables()
            if check:
                rng = np.random.Generator(np.random.PCG64(20261004))
                st = rng.bit_generator.state["state"]
                want = rng.normal(0, 1, 30000)
                got, used = restated_normals(st["state"], st["inc"], 30000, t)
                end = rng.bit_generator.state["state"]["state"]
                if not (np.array_equal(got, want) and pcg64_advance(st["state"], st["inc"], used) == end):
                    raise RuntimeError("numpy's Generator.normal is not the ziggurat fast_amd restates (numpy "
                                       f"{np.__version__}): GPU_RNG 'numpy' is unavailable")
            _TABLES = t
        return _TABLES


def state_words(bit_generator):
    """{state lo, state hi, inc lo, inc hi} of a PCG64 (the reference's default_rng), as the library takes them."""
    st = bit_generator.state
    if st.get("bit_generator") != "PCG64":
        raise RuntimeError(f"GPU_RNG 'numpy' restates PCG64 streams; the generator is {st.get('bit_generator')}")
    s, inc = st["state"]["state"], st["state"]["inc"]
    return np.array([s & M64, s >> 64, inc & M64, inc >> 64], dtype=np.uint64)


def set_state(bit_generator, lo_hi):
    """Put a PCG64 at the 128-bit state (lo, hi) the library returned.  Its increment is unchanged, and so is the 32-bit half-word
    an earlier integers() draw may have left buffered (has_uint32 / uinteger): Generator.normal() never reads or clears it, so
    the reference's generator would still hold it after the same draws."""
    st = bit_generator.state
    st["state"]["state"] = (int(lo_hi[1]) << 64) | int(lo_hi[0])
    bit_generator.state = st
