#!/usr/bin/env python3
"""Quick start: an uplink through an AO-corrected HV5/7 + Bufton atmosphere on one MI355X.

    python examples/quickstart.py            (after: python -c "import __graft_entry__ as g; g.build()")

Same configuration keys as ojdf/fast; `import fast` is an alias of `fast_amd` in this repository.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast                       # noqa: E402
from fast import comms            # noqa: E402

h, cn2, wind = fast.turbulence_models.HV57_Bufton_profile(4)
params = {
    "NPXLS": 1024, "DX": 0.01, "NITER": 100000, "NCHUNKS": 10, "SEED": 1, "LOGLEVEL": "ERROR",
    "D_GROUND": 0.8, "H_TURB": h, "CN2_TURB": cn2, "WIND_SPD": wind, "WIND_DIR": np.array([0., 90., 180., 270.]),
    "ZENITH_ANGLE": 55, "AO_MODE": "AO", "DSUBAP": 0.1, "ALIAS": True,
}
sim = fast.Fast(params)
res = sim.run()
print(res)
print(f"mean dB_rel {10 * np.log10(res._r.mean()):.2f}   scintillation index {res.scintillation_index:.4f}")
print(f"error budget [rad^2]: fitting {sim.fitting_error:.4f}  aniso-servo {sim.aniso_servo_error:.4f}  alias {sim.alias_error:.4f}")
t = sim.timing
print(f"GPU time of the run: rows {t['rows_ms']:.1f} ms, columns {t['cols_ms']:.1f} ms -> {params['NITER'] / (t['total_ms'] * 1e-3):.0f} iterations/s")
# reductions of the result vector where it already is, on the device
thr = 10 ** (-3 / 10) * res._r.mean()
print(f"P(fade below -3 dB of the mean) {comms.fade_prob(sim, thr):.4f}   OOK BER at Eb/N0 = 10 dB {comms.ber_ook(10.0, sim):.3e}"
      f"   16-QAM BER at 14 dB {comms.ber_qam(16, 14.0, sim):.3e}")
print("dB_rel histogram, 10 bins over [-10, 2] dB:", sim.histogram(-10.0, 2.0, 10)[:10])
