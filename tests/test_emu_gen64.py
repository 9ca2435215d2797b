"""The float64 device generator's FAST arithmetic (fast_amd/csrc/fmc_gen64.h: table-driven log, seeded cubic square root,
fdlibm kernels, integer quadrant logic) executed on the host by fast_amd/emu_gen64 and compared with the libm restatement of
the same definition (oracle/devrng.box_muller_f64).  Not a GPU test: everything in that header except the float32 1/sqrt
seed is plain IEEE float64 + FMA, which g++ reproduces exactly; tests/test_gpu_parity.py holds the device against the same
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
    """u at both ends and across every table interval boundary near 1 (the log's cancellation case), angles on and next
    to the quadrant boundaries and the rounding ties."""
    edge = []
    for a in (0, 1, 0xFFFFFFFF, 0xFFFFFFFE, 0x80000000, 0x7FFFFFFF, 0xC0000000, 0xBFFFFFFF, 0xFF800000, 0xFF7FFFFF, 0xFE000000, 0xFDFFFFFF):
        for a2 in (0, 0xFFFFFFFF, 0x7FF, 0x800, 0xFFFFF800, 0x400, 0x3FF):
            for b in (0, 0xFFFFFFFF, 0x20000000, 0x1FFFFFFF, 0x40000000, 0x3FFFFFFF, 0x60000000, 0x80000000, 0xA0000000, 0xC0000000,
                      0xE0000000, 0xDFFFFFFF, 0x1FFFFE00, 0x20000200):
                for b2 in (0, 0xFFFFFFFF, 3, 4):
                    edge.append((a, b, a2, b2))
    w = np.array(edge, dtype=np.uint32)
    got = run(emu, w, tmp_path)
    want = devrng.box_muller_f64(w[:, 0], w[:, 1], w[:, 2], w[:, 3])
    assert np.isfinite(got.view(np.float64)).all()
    assert np.abs(got - want).max() < 4e-15


def test_uniforms_next_to_one_keep_relative_accuracy(emu, tmp_path):
    """u -> 1: -2 ln u -> 0 and the normal is small; the table's two unit entries keep the RELATIVE error at rounding level
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
