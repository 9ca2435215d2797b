#!/usr/bin/env python3
"""Statistical battery on the DEVICE generator's own draws (GPU box): tools/generator_quality.py [realisations] [N] [seed]

Reads `realisations` x N^2 complex coefficients back through fastmc_rng_coeffs (the streams the row kernels consume) and
tests what the Monte-Carlo path relies on: normality (moments, tails to 6 sigma), exponential radius^2, uniform phase,
and independence -- between the two words of one xoshiro state advance (radius / angle of one coefficient), between
consecutive steps of a stream (kx, kx + streams per row), between neighbouring streams (lanes), rows, realisations and seeds.
Every statistic is printed as a z-score (|z| < 4.5 expected for all of them together)."""
import os
import sys
import numpy as np
from scipy import stats

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_amd import _lib   # noqa: E402
from oracle import devrng   # noqa: E402  (stream layout only; tools/ is test infrastructure)

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 20261003
h = _lib.Handle(N, 8, "f64", 0)
if len(sys.argv) > 4 and sys.argv[4] == "f32":      # default: the float64 generator (what a float64 handle draws since round 5)
    h.set_rng_precision("f32")
worst = 0.0


def report(name, z):
    global worst
    worst = max(worst, abs(z))
    print(f"{name:58s} z = {z:+.2f}")


def chi2_z(a, b, bins=64):
    H, _, _ = np.histogram2d(a, b, bins=bins, range=[[0, 1], [0, 1]])
    e = a.size / bins ** 2
    dof = bins ** 2 - 1
    return (((H - e) ** 2 / e).sum() - dof) / np.sqrt(2 * dof)


acc = dict(n=0, s1=0.0, s2=0.0, s3=0.0, s4=0.0)
tails = {k: 0 for k in (3.0, 4.0, 5.0, 6.0)}
pairs = {k: 0.0 for k in ("u-t same", "u-u step", "t-t step", "u-t step", "t-u step", "u-u lane", "u-u row")}
corr = {k: [0.0, 0] for k in ("re-re lane", "re-re step", "re-re row", "re-im", "re-re realisation", "re-re seed")}
hist_u = np.zeros(4096)
hist_t = np.zeros(4096)
prev = None
for g in range(R):
    c = h.rng_coeffs(SEED, g)
    z = np.concatenate([c.real.ravel(), c.imag.ravel()])
    acc["n"] += z.size
    for k, p in (("s1", 1), ("s2", 2), ("s3", 3), ("s4", 4)):
        acc[k] += (z ** p).sum()
    for k in tails:
        tails[k] += np.count_nonzero(np.abs(z) > k)
    u = np.exp(-np.abs(c) ** 2 / 2)
    t = (np.angle(c) + np.pi) / (2 * np.pi)
    hist_u += np.histogram(u, bins=4096, range=(0, 1))[0]
    hist_t += np.histogram(t, bins=4096, range=(0, 1))[0]
    S = devrng.stream_lanes(N)          # streams per row: 64, 128 / 256 at 2048 / 4096, 50 S on the 50-lane grids
    pairs["u-t same"] += chi2_z(u.ravel(), t.ravel())
    pairs["u-u step"] += chi2_z(u[:, :-S].ravel(), u[:, S:].ravel())
    pairs["t-t step"] += chi2_z(t[:, :-S].ravel(), t[:, S:].ravel())
    pairs["u-t step"] += chi2_z(u[:, :-S].ravel(), t[:, S:].ravel())
    pairs["t-u step"] += chi2_z(t[:, :-S].ravel(), u[:, S:].ravel())
    pairs["u-u lane"] += chi2_z(u[:, :-1].ravel(), u[:, 1:].ravel())
    pairs["u-u row"] += chi2_z(u[:-1].ravel(), u[1:].ravel())
    re, im = c.real, c.imag
    for k, (a, b) in (("re-re lane", (re[:, :-1], re[:, 1:])), ("re-re step", (re[:, :-S], re[:, S:])),
                      ("re-re row", (re[:-1], re[1:])), ("re-im", (re, im))):
        corr[k][0] += (a * b).sum()
        corr[k][1] += a.size
    if prev is not None:
        corr["re-re realisation"][0] += (re * prev).sum()
        corr["re-re realisation"][1] += re.size
    o = h.rng_coeffs(SEED + 1, g).real
    corr["re-re seed"][0] += (re * o).sum()
    corr["re-re seed"][1] += re.size
    prev = re

n = acc["n"]
m1, m2, m3, m4 = (acc[k] / n for k in ("s1", "s2", "s3", "s4"))
print(f"{n:.3e} normals from {R} realisations of {N}^2 complex coefficients, seed {SEED}")
report("mean", m1 * np.sqrt(n))
report("variance", (m2 - m1 ** 2 - 1) / np.sqrt(2 / n))
# standardised CENTRAL moments, whose standard errors under normality are sqrt(6 / n) and sqrt(24 / n).  (Rounds 2-5 divided the RAW
# third / fourth moments by these: the raw ones have variances 15 / n and 96 / n, so those z-scores read 1.58 x / 2 x too large --
# a "skewness z = -4.35" of round 6 was a raw-moment z of -2.75.)
c2 = m2 - m1 ** 2
c3 = m3 - 3 * m1 * m2 + 2 * m1 ** 3
c4 = m4 - 4 * m1 * m3 + 6 * m1 ** 2 * m2 - 3 * m1 ** 4
report("skewness", c3 / c2 ** 1.5 / np.sqrt(6 / n))
report("excess kurtosis", (c4 / c2 ** 2 - 3) / np.sqrt(24 / n))
for k, got in tails.items():
    e = n * 2 * stats.norm.sf(k)
    if e >= 30:
        zt = (got - e) / np.sqrt(e)
    else:       # few expected counts: the exact Poisson tail as a two-sided z (a count of 2 where 0.13 is expected is p = 0.8 %, not "5 sigma")
        pt = min(1.0, 2 * min(stats.poisson.cdf(got, e), stats.poisson.sf(got - 1, e)))
        zt = np.sign(got - e) * stats.norm.isf(pt / 2)
    report(f"count beyond {k:.0f} sigma: {got} (expected {e:.1f})", zt)
for name, hh in (("radius: exp(-|c|^2/2) uniform, 4096 bins", hist_u), ("phase uniform, 4096 bins", hist_t)):
    e = hh.sum() / 4096
    report(name, (((hh - e) ** 2 / e).sum() - 4095) / np.sqrt(2 * 4095))
for k, v in pairs.items():
    report(f"2-D chi^2 64x64, {k} (sum over realisations)", v / np.sqrt(R))
for k, (sxy, cnt) in corr.items():
    report(f"correlation {k}", sxy / np.sqrt(cnt))
print("worst |z| =", round(worst, 2), "OK" if worst < 4.5 else "CHECK")
