"""Host-side cost of one fastmc_run: a job too small to matter on the GPU (128^2, a few realisations) timed over many calls,
and the same for 5000 realisations (the 10 000-iteration step of tools/sizesweep.sh).  tools/call_overhead.py [calls]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_amd import _lib

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for N in (128, 256, 1024):
    Np = 82
    h = _lib.Handle(N, Np, "f64", 0)
    h.set_spectrum(np.full((N, N), 1e-4), 0.3)
    h.set_pupil(np.ones((Np, Np)), (N - Np) // 2, 0.01)
    for n in (8, 5000 if N < 1024 else 500):
        h.run(1, 0, n, None, 0.01)
        t0 = time.perf_counter()
        for i in range(calls if n == 8 else calls // 8):
            h.run(1, i * n, n, None, 0.01)
        dt = (time.perf_counter() - t0) / (calls if n == 8 else calls // 8)
        tim = h.last_timing()
        total = tim["total_ms"]; gpu = (tim["rows_ms"] + tim["cols_ms"] + tim.get("finalize_ms", 0.0)) if isinstance(tim, dict) else float("nan")
        print(f"N={N:5d} realisations per call {n:5d}: {dt * 1e6:8.1f} us per call, kernels {gpu * 1e3:8.1f} us, first launch to last event {total * 1e3:8.1f} us")
    h.close()
