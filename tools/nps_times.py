#!/usr/bin/env python3
"""Reads the per-tile timestamps a -DNPS1_EXP_TIMES build of the one-pass generator leaves (FASTMC_NPS_DUMP=file): start and
the 100 MHz-tick offsets of: generation done, entry offset known, index known, written.   python tools/nps_times.py FILE"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2)
lo, hi = raw[:, 0], raw[:, 1]
T0 = (lo & 0xffffffff).astype(np.int64)
d1 = ((lo >> 32) & 0xffffff).astype(np.int64)
d2 = (((lo >> 56) | (hi << 8)) & 0xffffff).astype(np.int64)
d3 = ((hi >> 16) & 0xffffff).astype(np.int64)
d4 = ((hi >> 40) & 0xffffff).astype(np.int64)
ok = (d4 < 100000) & (d4 > 0) & (np.abs(T0 - np.median(T0)) < 1e6)
s = T0[ok].min(); e = (T0 + d4)[ok].max()
print(f"{ok.sum()} tiles, span {(e - s) / 100:.1f} us")
for name, a in (("generation", d1), ("walk + entry", d2 - d1), ("index", d3 - d2), ("write", d4 - d3), ("tile", d4)):
    a = a[ok] / 100
    print(f"{name:14s} p50 {np.median(a):7.2f}  p90 {np.percentile(a, 90):7.2f}  max {a.max():7.2f} us")
