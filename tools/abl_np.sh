#!/bin/bash
# rows/cols ms per 5000 realisations for a given D_GROUND (window size): tools/abl_np.sh D lib...
D=$1; shift
for L in "$@"; do
  FASTMC_LIB=$PWD/$L DG=$D python - <<'PY'
import os, numpy as np, fast_amd, bench, argparse
a = argparse.Namespace(precision=os.environ.get("PREC", "f64"), npxls=int(os.environ.get("NPX", "1024")), ao_mode="NOAO", batch=0)
p = bench.workload_params(a); p["GPU_DEVICE"] = 0; p["D_GROUND"] = float(os.environ["DG"])
sim = fast_amd.Fast(p); h = sim._handle
for i in range(2): h.run(1, 0, 5000, None, float(sim.logamp_var), False)
t = {"rows_ms": 0, "cols_ms": 0}
for i in range(5):
    h.run(1, 0, 5000, None, float(sim.logamp_var), False)
    tt = h.last_timing(); t["rows_ms"] += tt["rows_ms"] / 5; t["cols_ms"] += tt["cols_ms"] / 5
print(os.path.basename(os.environ["FASTMC_LIB"]), a.npxls, "Np", sim.Npxls_pup, "rows %.3f ms  cols %.3f ms" % (t["rows_ms"], t["cols_ms"]))
PY
done
