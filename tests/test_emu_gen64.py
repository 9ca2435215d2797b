"""The float64 device generator's FAST arithmetic (fast_amd/csrc/fmc_gen64.h: table-driven log, seeded cubic square root,
table + rotation for the angle) executed on the host by fast_amd/emu_gen64 and compared with the libm restatement of
the same definition (oracle/devrng.box_muller_f64).  Not a GPU test: everything in that header except the float32 1/sqrt
seed is plain IEEE float64 + FMA, which g++ reproduces exactly; tests/test_gpu_parity_generator.py holds the device against the same
restatement (test_float64_device_generator_matches_its_restatement, bar 2e-14)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import devrng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "fast_amd", "emu_gen64")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-C", os.path.join(ROOT, "fast_amd", "csrc"), "emu-gen64"], check=True, capture_output=True)
    return EMU


def run(emu, words, tmp_path):
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    np.ascontiguousarray(words, dtype=np.uint32).tofile(fin)
    subprocess.run([emu, str(fin), str(fout)], check=True)
    got = np.fromfile(fout, dtype=np.float64).reshape(-1, 2)
    return got[:, 0] + 1j * got[:, 1]


def test_random_words_match_the_libm_restatement(emu, tmp_path):
    rng = np.random.default_rng(20261004)
    w = rng.integers(0, 2 ** 32, size=(400_000, 4), dtype=np.uint64).astype(np.uint32)
    got = run(emu, w, tmp_path)
    want = devrng.box_muller_f64(w[:, 0], w[:, 1], w[:, 2], w[:, 3])
    assert np.isfinite(got.view(np.float64)).all()
    d = np.abs(got - want)
    assert d.max() < 4e-15                                   # the GPU test's bar is 2e-14
    big = np.abs(want) > 1e-6
    assert (d[big] / np.abs(want[big])).max() < 1e-15


def test_edges_of_the_reductions(emu, tmp_path):
    """u at both ends (incl. the 64-bit integer that rounds up to 2^64: u = 1, a normal of exactly 0), across the carry of the
    [0.75, 1.5) reduction and every table interval boundary near 1 (the log's cancellation case), across the rounding of the 64-bit
    integer to 53 bits; angles on and next to the quadrant boundaries, the 256 table intervals' edges and centres."""
    edge = []
    for a in (0, 1, 0xFFFFFFFF, 0xFFFFFFFE, 0x80000000, 0x7FFFFFFF, 0xC0000000, 0xBFFFFFFF, 0xFF800000, 0xFF7FFFFF, 0xFE000000, 0xFDFFFFFF,
              0x00000400, 0x000003FF, 0x00200000, 0x001FFFFF):
        for a2 in (0, 0xFFFFFFFF, 0x3FF, 0x400, 0x401, 0xFFFFFC00, 0xFFFFFBFF, 0x7FF, 0x800, 0x80000000):
            for b in (0, 0xFFFFFFFF, 0x40000000, 0x3FFFFFFF, 0x80000000, 0x7FFFFFFF, 0xC0000000, 0xBFFFFFFF, 0x20000000, 0xE0000000,
                      0x01000000, 0x00FFFFFF, 0x00800000, 0x007FFFFF, 0xFF800000, 0x3F800000, 0x40800000):
                for b2 in (0, 0xFFFFFFFF, 1, 0x80000000, 0x7FFFFFFF):
                    edge.append((a, b, a2, b2))
    w = np.array(edge, dtype=np.uint32)
    got = run(emu, w, tmp_path)
    want = devrng.box_muller_f64(w[:, 0], w[:, 1], w[:, 2], w[:, 3])
    assert np.isfinite(got.view(np.float64)).all()
    assert np.abs(got - want).max() < 4e-15
    top = (w[:, 0] == 0xFFFFFFFF) & (w[:, 2] >= 0xFFFFFC00)          # a 2^32 + (a2 | 1) rounds to 2^64: u = 1
    assert top.any() and (got[top] == 0).all() and (want[top] == 0).all()


def test_uniforms_next_to_one_keep_relative_accuracy(emu, tmp_path):
    """u -> 1: -2 ln u -> 0 and the normal is small; the table's unit entry keeps the RELATIVE error at rounding level
    (a log reduced as k ln 2 + ln m with m in [1, 2) would lose it to cancellation)."""
    rng = np.random.default_rng(7)
    n = 100_000
    w = rng.integers(0, 2 ** 32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    sh = rng.integers(0, 32, size=n)
    w[: n // 2, 0] = 0xFFFFFFFF
    w[: n // 2, 2] = ((0xFFFFFFFF << sh[: n // 2]) & 0xFFFFFFFF).astype(np.uint32)
    w[n // 2:, 0] = ((0xFFFFFFFF << sh[n // 2:]) & 0xFFFFFFFF).astype(np.uint32)
    got = run(emu, w, tmp_path)
    want = devrng.box_muller_f64(w[:, 0], w[:, 1], w[:, 2], w[:, 3])
    ok = np.abs(want) > 1e-12
    assert (np.abs(got - want)[ok] / np.abs(want[ok])).max() < 2e-15
