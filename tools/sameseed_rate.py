#!/usr/bin/env python3
"""Same-seed modes at BASELINE configs[1] geometry (1024^2, Np 82, NOAO): `GPU_RNG: 'host'` (numpy draws on the host, uploads) against
`GPU_RNG: 'numpy'` (the same stream drawn on the device), same SEED, results compared.   python tools/sameseed_rate.py [NITER NCHUNKS]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, fast_amd
niter = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 100
h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
base = {"NPXLS": 1024, "DX": 0.01, "NCHUNKS": nchunks, "SEED": 1, "LOGLEVEL": "ERROR", "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w,
        "WIND_DIR": np.array([0., 90., 180., 270.]), "AO_MODE": "NOAO", "ZENITH_ANGLE": 55, "DSUBAP": 0.1, "GPU_DEVICE": 0}
# host mode on a sample (it is slow): the first 2 chunks' worth of iterations with the same chunking
m = niter // nchunks
ph = dict(base, GPU_RNG="host", NITER=2 * m, NCHUNKS=2)
sim = fast_amd.Fast(ph)
t0 = time.perf_counter(); rh = sim.run()._r; th = time.perf_counter() - t0
pn = dict(base, GPU_RNG="numpy", NITER=niter)
fast_amd.Fast(dict(pn, NITER=2 * m, NCHUNKS=2)).run()          # warm: tables, buffers, module load
sim = fast_amd.Fast(pn)
t0 = time.perf_counter(); rn = sim.run()._r; tn = time.perf_counter() - t0
tim = sim._handle.last_timing()
print(f"GPU_RNG 'host' : {2 * m / th:10.1f} it/s end to end ({2 * m} iterations)")
print(f"GPU_RNG 'numpy': {niter / tn:10.1f} it/s end to end ({niter} iterations in {nchunks} chunks, {tn * 1e3:.1f} ms)  -> {niter / tn / (2 * m / th):.0f} x")
# the logamp draws precede the chunks and depend on NITER, so compare like with like: a numpy-mode run of the 2-chunk job
sim2 = fast_amd.Fast(dict(pn, NITER=2 * m, NCHUNKS=2)); r2 = sim2.run()._r
print(f"same SEED, same job: max |device-draw / host-draw - 1| = {np.abs(r2 / rh - 1).max():.2e}")
