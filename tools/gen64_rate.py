#!/usr/bin/env python3
"""Rate of the float64 device generator at BASELINE configs[1] geometry: fused rows (MODE 2) vs the staged round-3 form
(FASTMC_GEN64_STAGED=1) vs the float32 generator; checks the fused draws against the read-back and the restatement.
    python tools/gen64_rate.py [N] [Np]"""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def one(N, Np, n_it=10000):
    from fast_amd import _lib
    rng = np.random.default_rng(3)
    fx = np.fft.fftshift(np.fft.fftfreq(N))
    k2 = fx[:, None] ** 2 + fx[None, :] ** 2
    ps = 1e-3 * (k2 + 1e-4) ** (-11 / 6)
    W = np.ones((Np, Np))
    out = {}
    for tag, prec in (("f32gen", "f32"), ("f64gen", "f64")):
        h = _lib.Handle(N, Np, "f64", 0)
        h.set_spectrum(ps, 1.0)
        h.set_pupil(W, (N - Np) // 2, 1.0)
        h.set_rng_precision(prec)
        h.run(1, 0, n_it // 2, None, 0.0, False)
        t0 = time.perf_counter()
        for i in range(3):
            r = h.run(1, (i + 1) * (n_it // 2), n_it // 2, None, 0.0, False)
        dt = time.perf_counter() - t0
        out[tag] = {"it_per_s": 3 * n_it / dt, "timing": h.last_timing() if hasattr(h, "last_timing") else None,
                    "kernel_path": h.kernel_path() if hasattr(h, "kernel_path") else None, "sample": [float(x) for x in r[:3]]}
    return out


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    Np = int(sys.argv[2]) if len(sys.argv) > 2 else 82
    if os.environ.get("GEN64_CHILD"):
        print(json.dumps(one(N, Np)))
    else:
        for env in ({}, {"FASTMC_GEN64_STAGED": "1"}):
            e = dict(os.environ, GEN64_CHILD="1", **env)
            r = subprocess.run([sys.executable, __file__, str(N), str(Np)], env=e, capture_output=True, text=True)
            print("staged" if env else "fused ", r.stdout.strip(), r.stderr.strip()[-400:])
