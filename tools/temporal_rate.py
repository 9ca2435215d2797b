"""Wall time of a TEMPORAL (frozen-flow) run, split into init and run: tools/temporal_rate.py [NITER]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fast_amd

niter = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
npx = int(sys.argv[2]) if len(sys.argv) > 2 else "auto"
h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
p = {"NPXLS": npx, "DX": 0.01, "NITER": niter, "NCHUNKS": 10, "TEMPORAL": True, "DT": 1e-3, "SEED": 1, "LOGLEVEL": "ERROR",
     "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": np.array([0., 90., 180., 270.]),
     "ZENITH_ANGLE": 55, "DSUBAP": 0.1, "AO_MODE": "AO", "ALIAS": True, "GPU_DEVICE": 0}
fast_amd.Fast(dict(p))                                   # first object: library load, caches
pi = cProfile.Profile(); pi.enable()
t0 = time.perf_counter(); sim = fast_amd.Fast(p); t1 = time.perf_counter()
pi.disable()
pstats.Stats(pi).sort_stats("tottime").print_stats(12)
pr = cProfile.Profile(); pr.enable()
r = sim.run()
pr.disable(); t2 = time.perf_counter()
print(f"N={sim.Npxls} Np={sim.Npxls_pup} NITER={niter}: init {t1 - t0:.2f} s, run {t2 - t1:.2f} s = {niter / (t2 - t1):.0f} steps/s; mean dB {10 * np.log10(r._r.mean()):.3f}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
