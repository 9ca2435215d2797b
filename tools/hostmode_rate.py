import sys, time; sys.path.insert(0, '.')
import numpy as np, fast_amd
h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
p = {"NPXLS": 1024, "DX": 0.01, "NITER": 200, "NCHUNKS": 10, "SEED": 1, "LOGLEVEL": "ERROR", "D_GROUND": 0.8, "H_TURB": h,
     "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": np.array([0., 90., 180., 270.]), "AO_MODE": "NOAO", "ZENITH_ANGLE": 55,
     "DSUBAP": 0.1, "GPU_RNG": "host", "GPU_DEVICE": 0}
sim = fast_amd.Fast(p)
t0 = time.perf_counter(); sim.run(); dt = time.perf_counter() - t0
rng = np.random.default_rng(0)
t0 = time.perf_counter(); rng.normal(size=(10, 1024, 1024)); rng.normal(size=(10, 1024, 1024)); tr = time.perf_counter() - t0
cr, ci = rng.normal(size=(10, 1024, 1024)), rng.normal(size=(10, 1024, 1024))
la = np.zeros(20)
t0 = time.perf_counter()
for _ in range(5): sim._handle.run_coeffs(cr, ci, la)
tg = (time.perf_counter() - t0) / 5
print(f"host-RNG mode: {200/dt:.1f} it/s end to end; numpy draw of one 20-iteration chunk {tr*1e3:.0f} ms; upload+GPU of that chunk {tg*1e3:.1f} ms -> {20/tg:.0f} it/s PCIe-inclusive")
