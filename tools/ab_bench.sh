#!/bin/bash
# A/B of library variants on ONE box: tools/ab_bench.sh <outdir> <reps> <name> [<name> ...]  (fast_amd/libfastmc_<name>.so;
# "main" = fast_amd/libfastmc.so).  Prints value / rows launch ms / cols ms per run; the JSON lines stay under <outdir>.
OUT=$1; REPS=$2; shift 2
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for n in "$@"; do
    LIB=$PWD/fast_amd/libfastmc_$n.so; [ "$n" == "main" ] && LIB=$PWD/fast_amd/libfastmc.so
    FASTMC_LIB=$LIB timeout 600 python bench.py --no-extras --no-cpu-baseline --no-sustained --no-f32-draw-pass --steps 10 ${AB_ARGS:-} > $OUT/bench_${n}_$rep.json 2> $OUT/bench_${n}_$rep.err
  done
done
python - "$OUT" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], 'value', round(d['value']), 'launch_ms', round(r['avg_launch_ms'],4), 'per', r['realisations_per_launch'], 'cols_ms', round(d['pipeline']['cols_ms'],3), r['kernel'])
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
