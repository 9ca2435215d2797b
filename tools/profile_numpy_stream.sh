#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats + two PMC passes of a same-seed run with numpy's stream drawn on the device
# (tools/sameseed_rate.py: 1024^2, 10 000 iterations in 100 chunks).  Output under gpurun_out/prof_<tag>/.
#   tools/profile_numpy_stream.sh <tag>
set -u
TAG=${1:-nps}; OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python3 $PWD/tools/sameseed_rate.py 10000 100"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_sq" -- $CMD > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $CMD > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
python3 - "$OUT" << 'PY' > "$OUT/summary.md"
import csv, glob, collections, sys
out = sys.argv[1]
print("# numpy's stream on the device: rocprofv3 of `tools/sameseed_rate.py 10000 100` (1024^2, 100 chunks of 50 realisations)\n")
for line in open(out + "/trace.log"):
    if "it/s" in line or "same SEED" in line: print("    " + line.rstrip())
print("\n## kernel stats\n\n| kernel | calls | avg us | max us | % |\n|---|---|---|---|---|")
f = sorted(glob.glob(out + "/trace/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {r['Percentage']} |")
med = lambda x: sorted(x)[len(x) // 2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{out}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n## counters (median per dispatch over the full-size dispatches)\n")
for k, v in acc.items():
    if "onepass" not in k and "k_rows_wave" not in k: continue
    c = {n: med([y for y in x if y >= 0.5 * max(x)]) for n, x in v.items()}
    print(f"### {k[:80]}\n")
    for n in sorted(c): print(f"* {n}: {c[n]:.4g}")
    if "GRBM_GUI_ACTIVE" in c and "onepass" in k:      # (GRBM_GUI_ACTIVE of the short row dispatches includes the gaps between the serialised dispatches)
        simd = c["GRBM_GUI_ACTIVE"] / 8 * 1024
        print(f"* busy fractions of the SIMD cycles (SQ_ACTIVE_INST_* x 4 / (GRBM_GUI_ACTIVE / 8 x 1024)): VALU {c['SQ_ACTIVE_INST_VALU'] * 4 / simd:.2f}, "
              f"scalar {c['SQ_ACTIVE_INST_SCA'] * 4 / simd:.2f}, LDS {c['SQ_ACTIVE_INST_LDS'] * 4 / simd:.2f}, any {c['SQ_ACTIVE_INST_ANY'] * 4 / simd:.2f}")
    if "FETCH_SIZE" in c: print(f"* HBM per dispatch: fetch {c['FETCH_SIZE'] * 1024 / 1e6:.0f} MB (FETCH_SIZE in KB; x 2 on gfx950 per the guide: {c['FETCH_SIZE'] * 2048 / 1e6:.0f} MB), write {c.get('WRITE_SIZE', 0) * 1024 / 1e6:.0f} MB")
    print()
PY
rm -rf "$OUT"/*/*/*.db 2>/dev/null
cat "$OUT/summary.md"
