"""Differential fuzz: wave / chirp-z / 50-lane family vs direct family on random (N, Np, lo, precision) with host coefficients
(screens) and with the device generator (powers), and the screens against numpy for the smaller grids (odd N use
numpy's asymmetric fftshift).  tools/fuzz_families.py [cases] [seed] [N1,N2,...: only these grid sizes]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_amd import _lib, host

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
bad = 0
for c in range(cases):
    N = int(rng.choice(host.WAVE_FFT_SIZES))
    if N >= 2048 and rng.random() < 0.5:
        N = int(rng.choice([s for s in host.WAVE_FFT_SIZES if s < 2048]))
    if c % 2:                                           # every other case: a size that is not 64 P -> chirp-z family
        N = int(rng.integers(8, 1500))
    if c % 16 == 5:                                     # now and then a grid beyond one chirp-z transform: rows in input blocks
        N = int(rng.integers(1968, 4096))
    if c % 4 == 3:                                      # every fourth: N = 50 P -> 50-lane family
        N = int(rng.choice([100, 150, 200, 250, 300, 350, 400, 450, 500, 600, 700, 800, 900, 1000, 1200, 1350, 1400, 1500, 1600, 1750, 1800, 2000,
                          1344, 1728, 1920, 2304, 2560]))
    if c % 32 == 9:                                     # rarely a grid beyond 4096: sub-rows (64 P S, 50 P S) or chirp-z blocks
        N = int(rng.choice([4608, 5000, 5120, 6000, 6144, 6400, 7000, 7168, 7680, 8000, 8192, int(rng.integers(4097, 8192)), int(rng.integers(4097, 8192))]))
    if c % 8 == 6:                                      # any multiple of 64: the packed sub-rows with a run-time count (fmc_core.h: pks_rt) among them
        N = 64 * int(rng.integers(3, 64))
    if only:
        N = int(rng.choice(only))
    Np = int(rng.integers(1, min(N, 300 if c % 2 == 0 else 256) + 1))
    if N > 2048 and c % 16 == 5:
        Np = int(rng.integers(1, 200))
    if N > 4096:
        Np = int(rng.integers(1, 200))
    lo = int(rng.choice([0, N - Np, (N - Np) // 2, rng.integers(0, N - Np + 1)]))
    if N % 64 == 0 and 192 <= N <= 8192 and N not in (256, 512, 1024) and rng.random() < 0.6:
        # grids of the packed sub-rows (round 6): two cases in three inside the 96 outputs their six planes hold
        span = 48 if rng.random() < 0.5 else (64 if rng.random() < 0.6 or N % 256 else 128)      # six planes; eight (windows of 97 ... 128 pixels); all sixteen (... 256, N = S x 256)
        Np = int(rng.integers(1, 2 * span + 1))
        lo = int(rng.integers(max(N // 2 - span, 0), N // 2 + span - Np + 1))
    prec = "f64" if rng.random() < 0.7 else "f32"
    tol = 1e-10 if prec == "f64" else 1e-4
    ps = rng.uniform(0.0, 1.0, size=(N, N)) ** 4 * 1e-3
    cr, ci = rng.normal(size=(1, N, N)), rng.normal(size=(1, N, N))
    h = _lib.Handle(N, Np, prec, 0)
    h.set_rng_precision("f32")                       # the float32 draw first (fastmc_create leaves a float64 handle at the float64 generator)
    h.set_spectrum(ps, 0.37)
    h.set_pupil(np.ones((Np, Np)), lo, 0.01)
    a = h.screens_coeffs(cr, ci)
    ra = h.run(c + 1, 3, 2, None, 0.01)              # device generator, detector, finalize
    rows_kernel = h.last_kernels()[0].split("<")[0]
    g64 = prec == "f64" and os.environ.get("FUZZ_GEN64", "1") != "0"
    if g64:                                          # ... and the float64 generator, fused into the family's rows (MODE 2) where it has the form
        h.set_rng_precision("f64")
        ra64 = h.run(c + 1, 3, 2, None, 0.01)
        h.set_rng_precision("f32")
    path = h.kernel_path()
    h.kernel_path(0)
    b = h.screens_coeffs(cr, ci)
    rb = h.run(c + 1, 3, 2, None, 0.01)
    err = max(np.abs(a - b).max() / np.abs(b).max(), np.abs(ra - rb).max() / np.abs(rb).max() * (1e-2 if prec == "f32" else 1e-1))
    if g64:                                          # the direct family stages the same draws (k_gen_coeffs_f64): float64 end to end
        h.set_rng_precision("f64")
        rb64 = h.run(c + 1, 3, 2, None, 0.01)
        err = max(err, np.abs(ra64 - rb64).max() / np.abs(rb64).max() * 1e-1)
    h.close()
    ref_err = float("nan")
    if N <= 1024:
        z = np.fft.fftshift(np.fft.fft2(np.fft.fftshift((cr[0] + 1j * ci[0]) * np.sqrt(ps) * 0.37)))[lo:lo + Np, lo:lo + Np]
        ref_err = max(np.abs(a[0] - z.real).max(), np.abs(a[1] - z.imag).max()) / np.abs(z).max()
    ok = err < tol and not (ref_err >= tol)
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} N={N:5d} Np={Np:4d} lo={lo:5d} {prec} path={path} {rows_kernel} vs-direct {err:.2e} vs-numpy {ref_err:.2e}")
print("failures:", bad)
sys.exit(1 if bad else 0)
