#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats output directory: tools/kernel_stats_top.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s}  total {float(r['TotalDurationNs']) / 1e6:9.3f} ms  {r['Percentage']:>6s} %")
