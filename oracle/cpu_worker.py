"""One process of the all-cores CPU baseline of bench.py (test infrastructure, like the rest of
oracle/): `python -m oracle.cpu_worker inputs.npz seed chunks` runs `chunks` chunks of 20
iterations of oracle.fastref.monte_carlo and prints `iterations seconds`.  Started as a child
program (never forked from the GPU-initialised bench process)."""
import os
import sys
import time

for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "1")

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fastref as R  # noqa: E402


def main():
    path, seed, chunks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    g = np.load(path)
    t0 = time.perf_counter()
    r = R.monte_carlo(seed, 20 * chunks, chunks, g["ps"], float(g["df"]), g["W"], float(g["dx"]), float(g["lv"]))
    dt = time.perf_counter() - t0
    assert np.isfinite(r).all()
    print(20 * chunks, dt)


if __name__ == "__main__":
    main()
