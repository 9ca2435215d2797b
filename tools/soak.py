import sys, gc; sys.path.insert(0, '.')
import numpy as np, fast_amd, ctypes
hip = ctypes.CDLL("libamdhip64.so")
def free_mem():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value / 2**20
h, cn2, w = fast_amd.turbulence_models.HV57_Bufton_profile(4)
p = {"NPXLS": 256, "DX": 0.01, "NITER": 2000, "NCHUNKS": 10, "SEED": 1, "LOGLEVEL": "ERROR", "D_GROUND": 0.8, "H_TURB": h,
     "CN2_TURB": cn2, "WIND_SPD": w, "WIND_DIR": np.array([0., 90., 180., 270.]), "DSUBAP": 0.1, "GPU_DEVICE": 0, "SUBHARM": True}
m0 = None
for i in range(150):
    p["SEED"] = i; p["GPU_PRECISION"] = "f32" if i % 3 == 0 else "f64"
    # every fifth object draws numpy's stream on the device (one-pass generator and its buffers), every seventh the opt-in float32 draw,
    # now and then another grid family (50-lane, chirp-z)
    p["GPU_RNG"] = "numpy" if (i % 5 == 1 and p["GPU_PRECISION"] == "f64") else "device"
    p["GPU_RNG_PRECISION"] = "f32" if i % 7 == 2 else "auto"     # every seventh the opt-in float32 draw, else the default
    p["NPXLS"] = (256, 256, 300, 256, 291)[i % 5] if p["GPU_PRECISION"] == "f64" else 256
    sim = fast_amd.Fast(dict(p)); r = sim.run()._r; st = sim.result_stats([-3.0]); hs = sim.histogram()
    assert np.isfinite(r).all() and hs.sum() == 2000
    del sim
    if i == 10: gc.collect(); m0 = free_mem()
gc.collect(); m1 = free_mem()
print(f"free device memory after 10 objects {m0:.0f} MiB, after 150 objects {m1:.0f} MiB, drift {m0-m1:.1f} MiB")
