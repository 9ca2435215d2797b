"""CPU: numpy's Generator.normal stream as fast_amd restates it for `GPU_RNG: 'numpy'` (fast_amd/npnormal.py): the ziggurat
tables are read out of the installed numpy through a crafted bit generator, and the restated stream -- PCG64, fast path, wedge
test, tail loop, restarts -- is numpy's own, value for value and word for word.  The device implementation
(fast_amd/csrc/fmc_npstream.h) is held to the same stream under `-m gpu` (tests/test_gpu_npstream.py)."""
import numpy as np
import pytest

from fast_amd import npnormal


def test_tables_come_out_of_numpy_and_have_the_ziggurat_structure():
    wi, ki, fi = npnormal.get_tables()
    x = wi * 2.0 ** 52                               # layer edges
    assert ki[1] == 0 and (ki[2:] > 0).all() and (ki < 2 ** 52).all()
    assert abs(x[255] - npnormal.ZIG_R) < 1e-12 and (np.diff(x[1:]) > 0).all() and x[0] > x[255]
    assert fi[0] == 1.0 and (np.diff(fi) < 0).all()
    # ki[i] = floor(2^52 x_{i-1} / x_i) for the proper layers (Marsaglia & Tsang)
    assert np.abs(ki[2:].astype(float) / 2.0 ** 52 - x[1:-1] / x[2:]).max() < 1e-12


@pytest.mark.parametrize("seed", [0, 1, 2, 20261004, 2 ** 63 + 5])
def test_restated_stream_is_numpys(seed):
    rng = np.random.default_rng(seed)
    st = rng.bit_generator.state["state"]
    n = 120_000
    want = rng.normal(0, 1, n)
    got, used = npnormal.restated_normals(st["state"], st["inc"], n)
    assert np.array_equal(got, want)
    assert used > n * 1.015                          # the slow paths were exercised (2.2 % extra words)
    assert npnormal.pcg64_advance(st["state"], st["inc"], used) == rng.bit_generator.state["state"]["state"]
    # two consecutive arrays of one stream are one array (how fast.py:639-645 and funcs.py:352-356 draw)
    rng2 = np.random.default_rng(seed)
    assert np.array_equal(np.concatenate([rng2.normal(0, 1, 1000), rng2.normal(0, 1, (3, 50, 2)).ravel()]), want[:1300])


def test_state_round_trip():
    rng = np.random.default_rng(9)
    w = npnormal.state_words(rng.bit_generator)
    st = rng.bit_generator.state["state"]
    new = npnormal.pcg64_advance(st["state"], st["inc"], 12345)
    npnormal.set_state(rng.bit_generator, (new & npnormal.M64, new >> 64))
    ref = np.random.default_rng(9)
    ref.bit_generator.advance(12345)
    assert np.array_equal(rng.normal(size=5), ref.normal(size=5)) and int(w[0]) == st["state"] & npnormal.M64
    with pytest.raises(RuntimeError):
        npnormal.state_words(np.random.Generator(np.random.MT19937(1)).bit_generator)


def test_set_state_keeps_a_buffered_half_word():
    """Generator.normal() never touches the 32-bit half-word an earlier integers() draw left buffered, so writing the
    state the device returned back must not clear it either -- the stream after run() is then the reference's."""
    from fast_amd import npnormal
    ref = np.random.default_rng(5)
    ref.integers(0, 10, dtype=np.uint32)             # buffers the other half of a 64-bit word
    assert ref.bit_generator.state["has_uint32"] == 1
    ours = np.random.default_rng(5)
    ours.integers(0, 10, dtype=np.uint32)
    ref.normal(size=1000)                            # the reference draws ...
    sw = npnormal.state_words(ref.bit_generator)     # ... and where it stands is what the device reports for the same draws
    npnormal.set_state(ours.bit_generator, sw[:2])
    assert ours.bit_generator.state == ref.bit_generator.state
    assert np.array_equal(ours.integers(0, 2 ** 32, size=5, dtype=np.uint32), ref.integers(0, 2 ** 32, size=5, dtype=np.uint32))
