#!/usr/bin/env python3
"""Kernel times of the numpy-stream generator for one array of N normals (run under rocprofv3 --kernel-trace --stats).
   python tools/nps_kernel_time.py [N]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from fast_amd import _lib, npnormal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 209_715_200
h = _lib.Handle(256, 40, "f64", 0)
rng = np.random.default_rng(1)
sw = npnormal.state_words(rng.bit_generator)
for _ in range(4):
    got, after, consumed, ovf = h.npstream_normals(sw, n)
print("overflow flags", ovf, "consumed / n", consumed / n)
