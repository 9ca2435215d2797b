#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times (rocprofv3 --kernel-trace --stats) of the bench step at several grid sizes, float64 generator,
# one step at a time (--no-pipeline), and the SQ counter passes for the sizes named after "--pmc".
#   tools/trace_sizes.sh <tag> <sizes...> [--pmc <sizes...>]
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PMC=0
for n in "$@"; do
  if [ "$n" == "--pmc" ]; then PMC=1; continue; fi
  BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-sustained --no-f32-draw-pass --no-host-cost-pass --no-pipeline --rng-precision f64 --npxls $n"
  if [ $PMC == 0 ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t$n" -- $BENCH > "$OUT/t$n.log" 2>&1
    echo "== N = $n: $(grep '^{"metric"' $OUT/t$n.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), "it/s under rocprofv3")')"
    python3 - "$OUT/t$n" <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print(f"  {r['Name'].replace('void fmc::','').split('(')[0][:58]:58s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {float(r['Percentage']):5.1f} %")
PY
  else
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d "$OUT/p1_$n" -- $BENCH > "$OUT/p1_$n.log" 2>&1
    rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p2_$n" -- $BENCH > "$OUT/p2_$n.log" 2>&1
    echo "== N = $n counters (per-dispatch averages of the row kernel)"
    python3 - "$OUT" $n <<'PY'
import csv, glob, sys, os, collections
out, n = sys.argv[1], sys.argv[2]
for sub in ("p1_", "p2_"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in glob.glob(os.path.join(out, sub + n, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void fmc::", "").split("(")[0]
            if k.startswith("k_rows"):
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in agg:
        print("  ", k, {c: round(v / len(cnt[k])) for c, v in sorted(agg[k].items())})
PY
  fi
done
rm -rf "$OUT"/*/*/*.db 2>/dev/null
