#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile_gpu.sh) into one markdown summary."""
import csv, glob, os, sys, collections

out = sys.argv[1]

def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))

def short(name):
    n = name.replace("void fmc::", "").replace("fmc::", "")
    return n.split("(")[0][:60]

print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
for f in find("trace", "*kernel_stats.csv"):
    print("## kernel stats (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
              f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    print()
for f in find("trace", "*kernel_trace.csv")[:1]:
    rows = list(csv.DictReader(open(f)))
    seen = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        if k not in seen:
            seen[k] = r
    print("## launch geometry / resources (first dispatch of each kernel)\n")
    print("| kernel | grid | workgroup | VGPR | accum VGPR | SGPR | LDS B | scratch B |")
    print("|---|---|---|---|---|---|---|---|")
    for k, r in seen.items():
        print(f"| {k} | {r.get('Grid_Size_X','?')} | {r.get('Workgroup_Size_X','?')} | {r.get('VGPR_Count','?')} | "
              f"{r.get('Accum_VGPR_Count','?')} | {r.get('SGPR_Count','?')} | {r.get('LDS_Block_Size','?')} | {r.get('Scratch_Size','?')} |")
    print()
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_sq4"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    print(f"## PMC pass {sub} (per-dispatch averages)\n")
    names = sorted({c for k in agg for c in agg[k]})
    print("| kernel | dispatches | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for k in agg:
        n = max(len(cnt[k]), 1)
        print(f"| {k} | {n} | " + " | ".join(f"{agg[k][c]/n:.4g}" for c in names) + " |")
    print()
