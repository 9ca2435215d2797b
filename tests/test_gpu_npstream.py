"""GPU: numpy's Generator.normal stream drawn on the device (fast_amd/csrc/fmc_npstream.h) against numpy itself -- the oracle
here is the library the reference calls (fast/funcs.py:21, 352-365)."""
import numpy as np
import pytest

import fast_amd
from fast_amd import _lib, npnormal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n", [(1, 1), (1, 7), (2, 5000), (3, 16384), (4, 16385), (5, 100_000), (6, 1_048_576), (2 ** 40 + 7, 3_000_001)])
def test_device_array_is_numpys(seed, n):
    h = _lib.Handle(256, 40, "f64", 0)
    rng = np.random.default_rng(seed)
    sw = npnormal.state_words(rng.bit_generator)
    want = rng.normal(0, 1, n)
    got, after, consumed, ovf = h.npstream_normals(sw, n)
    assert ovf == 0
    d = np.flatnonzero(got != want)
    # ocml's log1p / exp against glibc's: a tail or wedge VALUE is the same number (the fast path and the wedge return rabs * wi),
    # only the tail's R + xx can differ in the last bit
    assert d.size <= max(2, n // 200_000) and np.abs(got - want).max() < 1e-15 * 8
    end = rng.bit_generator.state["state"]["state"]
    assert (int(after[1]) << 64) | int(after[0]) == end                  # the stream position is numpy's, word for word
    st0 = {"lo": int(sw[0]), "hi": int(sw[1])}
    assert npnormal.pcg64_advance((st0["hi"] << 64) | st0["lo"], (int(sw[3]) << 64) | int(sw[2]), consumed) == end
    assert n <= consumed < 1.03 * n + 40


def test_consecutive_arrays_continue_the_stream():
    h = _lib.Handle(256, 40, "f64", 0)
    rng = np.random.default_rng(77)
    sw = npnormal.state_words(rng.bit_generator)
    for n in (1000, 20000, 333):
        want = rng.normal(0, 1, n)
        got, after, _, ovf = h.npstream_normals(sw, n)
        assert ovf == 0 and np.abs(got - want).max() < 1e-14
        sw = np.array([after[0], after[1], sw[2], sw[3]], dtype=np.uint64)
