#!/usr/bin/env python3
"""Iterations/s of the device generator over window sizes Np (centred) at grid size N: where the packed forms end (96 pixels on the
multiples of 64, 128 on the chirp-z grids) the float64 generator is staged through memory onto the one-row-per-wave rows.
    python tools/window_rate.py N Np1 [Np2 ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fast_amd import _lib

N = int(sys.argv[1])
fx = np.fft.fftshift(np.fft.fftfreq(N))
ps = 1e-3 * (fx[:, None] ** 2 + fx[None, :] ** 2 + 1e-4) ** (-11 / 6)
n_it = max(400, int(10000 * (1024.0 / N) ** 2))
for Np in [int(x) for x in sys.argv[2:]]:
    h = _lib.Handle(N, Np, "f64", 0)
    h.set_spectrum(ps, 1.0)
    h.set_pupil(np.ones((Np, Np)), (N - Np) // 2, 1.0)
    h.run(1, 0, n_it // 2, None, 0.0, False)
    t0 = time.perf_counter()
    for i in range(3):
        h.run(1, (i + 1) * (n_it // 2), n_it // 2, None, 0.0, False)
    dt = time.perf_counter() - t0
    print(f"N={N} Np={Np:4d} {3 * n_it / dt:12.0f} it/s  rows {h.last_kernels()[0]}  cols {h.last_kernels()[1]}", flush=True)
    h.close()
