#!/usr/bin/env python3
"""Design of the fast float64 Box-Muller of fast_amd/csrc/fmc_gen64.h (build-time helper, not product code).

The float64 device generator (GPU_RNG_PRECISION 'f64', round-5 definition) needs  y = -2 ln u  for
u = RNE(a 2^32 + (a2 | 1)) 2^-64.  The kernel reduces u = 2^K m, m in [0.75, 1.5), looks (-2 c_j, 2 ln c_j) up in a 128-entry
table indexed by the top seven mantissa bits of the reduced hi word, forms r' = -2 (m c_j - 1) with one FMA and evaluates

    -2 ln(1 + r) = r' + r'^2 Q(r'),     r = -r'/2,   |r'| <= 2^-7 (1 + 2^-7)

(c_j = 1 only for the interval that ends at m = 1: there K = 0 and y = p exactly, so u -> 1 keeps full RELATIVE accuracy; the
interval that starts at m = 1 only occurs with K <= -1, y >= 0.57, and takes an ordinary c_j -- round 4 gave it c = 1 as well,
which doubled the range of r' and cost Q a degree.)

This script finds the range of r', fits Q (near-minimax, Chebyshev nodes, mpmath at 60 digits), prints the coefficients as C
hex-float literals with the error bound, prints the scaled Taylor coefficients of the angle's rotation, and checks the whole
reduction against mpmath on random inputs.  Run:  python tools/gen64_design.py
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60
DEG = 4


def Q_exact(x):
    x = mp.mpf(x)
    if abs(x) < mp.mpf(2) ** -40:
        return mp.mpf(1) / 4 + x / 12 + x * x / 32
    return (-2 * mp.log(1 - x / 2) - x) / (x * x)


def fit(deg, a, b):
    poly, err = mp.chebyfit(Q_exact, [a, b], deg + 1, error=True)
    return [mp.mpf(c) for c in poly][::-1], err      # ascending powers


def f64_from_hi(hi):
    return float(np.uint64(int(hi) << 32).view(np.float64))


def table_c(j):
    """c_j: reciprocal of the centre of interval j rounded to float32; exactly 1 for the interval [1 - 2^-8, 1)."""
    if j == 63:
        return 1.0
    lo = f64_from_hi((j << 13) + 0x3FE80000)
    hi = f64_from_hi(((j + 1) << 13) + 0x3FE80000)
    return float(np.float32(1.0 / (0.5 * (lo + hi))))


def r_range():
    lo_all, hi_all = 0, 0
    for j in range(128):
        c = mp.mpf(table_c(j))
        m0 = mp.mpf(f64_from_hi((j << 13) + 0x3FE80000))
        m1 = mp.mpf(f64_from_hi(((j + 1) << 13) + 0x3FE80000))
        ra, rb = -2 * (m0 * c - 1), -2 * (m1 * c - 1)
        lo_all, hi_all = min(lo_all, ra, rb), max(hi_all, ra, rb)
    return lo_all, hi_all


def main():
    a, b = r_range()
    print(f"r' in [{mp.nstr(a, 8)}, {mp.nstr(b, 8)}]  (2^-7 = {2.0 ** -7})")
    for deg in (3, 4, 5):
        co, err = fit(deg, a, b)
        print(f"deg {deg}: max |dQ| = {mp.nstr(err, 5)}  -> relative error of y <= {mp.nstr(err * max(-a, b), 5)}")
    co, err = fit(DEG, a, b)
    print("// Q(r') coefficients, ascending (tools/gen64_design.py):")
    for i, c in enumerate(co):
        print(f"  {float(c).hex()},   // q{i} = {mp.nstr(c, 20)}")
    # the rotation by x = k w, k = 2 pi 2^-56, |x| <= pi / 256:  sin x = w (k1 + z (k3 + z k5)),  cos x - 1 = z (c2 + z (c4 + z c6)), z = w^2
    k = 2 * mp.pi * mp.mpf(2) ** -56
    print("// angle: scaled Taylor coefficients (x = k w, w = G - 2^47, k = 2 pi 2^-56)")
    for name, v in (("k1", k), ("k3", -k ** 3 / 6), ("k5", k ** 5 / 120), ("c2", -k ** 2 / 2), ("c4", k ** 4 / 24), ("c6", -k ** 6 / 720)):
        print(f"  {name} = {float(v).hex()}   // {mp.nstr(v, 20)}")
    x = mp.pi / 256
    print(f"// truncation at |x| = pi/256: sin {mp.nstr(x ** 7 / 5040, 3)}, cos {mp.nstr(x ** 8 / 40320, 3)}")
    # end-to-end check of the reduction in exact arithmetic with the rounded coefficients
    q = [mp.mpf(float(c)) for c in co]
    rng = np.random.default_rng(1)
    worst = 0
    for _ in range(4000):
        A = int(rng.integers(0, 2 ** 63)) * 2 + 1
        if rng.random() < 0.3:
            A = (2 ** 64 - 1 - int(rng.integers(0, 2 ** int(rng.integers(1, 60))))) | 1     # u close to 1
        v = np.float64(A)                       # RNE of the 64-bit odd integer
        bits = int(v.view(np.uint64))
        hx = (bits >> 32) + 0x80000
        K = (hx >> 20) - (1023 + 64)
        mhi = (hx & 0xFFFFF) + 0x3FE80000
        m = np.uint64((mhi << 32) | (bits & 0xFFFFFFFF)).view(np.float64)
        j = (hx >> 13) & 0x7F
        c = table_c(j)
        rp = -2 * (mp.mpf(float(m)) * mp.mpf(c) - 1)
        assert a <= rp <= b, (j, rp)
        y = K * (-2 * mp.log(2)) + 2 * mp.log(mp.mpf(c)) + rp + rp * rp * sum(qc * rp ** i for i, qc in enumerate(q))
        ref = -2 * mp.log(mp.mpf(float(v)) * mp.mpf(2) ** -64)
        if ref != 0:
            worst = max(worst, abs(y / ref - 1))
    print("reduction + polynomial vs mpmath: max relative error of y =", mp.nstr(worst, 5))


if __name__ == "__main__":
    main()
