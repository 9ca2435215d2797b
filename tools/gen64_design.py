#!/usr/bin/env python3
"""Design of the fast float64 Box-Muller of fast_amd/csrc/fmc_gen64.h (build-time helper, not product code).

The float64 device generator (GPU_RNG_PRECISION 'f64') needs  y = -2 ln u  for a 53-bit uniform u.  The kernel reduces
u = 2^K m, m in [0.75, 1.5), looks (c_j, 2 ln c_j) up in a 128-entry table indexed by the top seven mantissa bits of the
reduced hi word, forms r' = -2 (m c_j - 1) with one FMA and evaluates

    -2 ln(1 + r) = r' + r'^2 Q(r'),     r = -r'/2,   r' in [-2^-6, 2^-7]

This script fits Q (near-minimax, Chebyshev nodes, mpmath at 60 digits), prints the coefficients as C hex-float literals and
the error bound, and checks the whole reduction against mpmath on random inputs.  Run:  python tools/gen64_design.py
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def Q_exact(x):
    x = mp.mpf(x)
    if abs(x) < mp.mpf(2) ** -40:
        return mp.mpf(1) / 4 + x / 12 + x * x / 32
    return (-2 * mp.log(1 - x / 2) - x) / (x * x)


def fit(deg, a, b):
    poly, err = mp.chebyfit(Q_exact, [a, b], deg + 1, error=True)
    return [mp.mpf(c) for c in poly][::-1], err      # ascending powers


def main():
    a, b = -mp.mpf(2) ** -6, mp.mpf(2) ** -7
    for deg in (4, 5, 6):
        co, err = fit(deg, a, b)
        print(f"deg {deg}: max |dQ| = {mp.nstr(err, 5)}  -> relative error of y <= {mp.nstr(err * 2 ** -6, 5)}")
    co, err = fit(5, a, b)
    print("// Q(r') coefficients, ascending (tools/gen64_design.py):")
    for i, c in enumerate(co):
        print(f"  {float(c).hex()},   // q{i} = {mp.nstr(c, 20)}")
    # end-to-end check of the reduction in exact arithmetic with the rounded coefficients
    q = [mp.mpf(float(c)) for c in co]
    rng = np.random.default_rng(1)
    worst = 0
    for _ in range(4000):
        A = int(rng.integers(0, 2 ** 53))
        if rng.random() < 0.3:
            A = 2 ** 53 - 1 - int(rng.integers(0, 2 ** int(rng.integers(1, 50))))     # u close to 1
        v = float(A) + 0.5
        hi = np.float64(v).view(np.uint64) >> np.uint64(32)
        hx = int(hi) + 0x80000
        K = (hx >> 20) - 1076
        mhi = (hx & 0xFFFFF) + 0x3FE80000
        m = np.uint64((mhi << 32) | (int(np.float64(v).view(np.uint64)) & 0xFFFFFFFF)).view(np.float64)
        j = (hx >> 13) & 0x7F
        c = table_c(j)
        rp = -2 * (mp.mpf(float(m)) * mp.mpf(c) - 1)
        y = K * (-2 * mp.log(2)) + 2 * mp.log(mp.mpf(c)) + rp + rp * rp * sum(qc * rp ** i for i, qc in enumerate(q))
        ref = -2 * mp.log(mp.mpf(v) * mp.mpf(2) ** -53)
        if ref != 0:
            worst = max(worst, abs(y / ref - 1))
    print("reduction + polynomial vs mpmath: max relative error of y =", mp.nstr(worst, 5))


def table_c(j):
    """c_j: reciprocal of the centre of interval j rounded to float32 (1 for the two intervals that touch m = 1)."""
    if j in (63, 64):
        return 1.0
    lo = np.uint64(((j << 13) + 0x3FE80000) << 32).view(np.float64)
    hi = np.uint64((((j + 1) << 13) + 0x3FE80000) << 32).view(np.float64)
    return float(np.float32(1.0 / (0.5 * (float(lo) + float(hi)))))


if __name__ == "__main__":
    main()
