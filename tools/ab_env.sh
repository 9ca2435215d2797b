#!/bin/bash
# A/B of ONE environment switch on one box: tools/ab_env.sh <outdir> <reps> <VAR> <value> [<value> ...]  ("-" = unset).
# Same report as tools/ab_bench.sh; extra bench.py arguments in AB_ARGS.
OUT=$1; REPS=$2; VAR=$3; shift 3
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    if [ "$v" == "-" ]; then unset $VAR; else export $VAR=$v; fi
    timeout 600 python bench.py --no-extras --no-cpu-baseline --no-sustained --no-f32-draw-pass --steps 10 ${AB_ARGS:-} > $OUT/bench_${VAR}_${v}_$rep.json 2> $OUT/bench_${VAR}_${v}_$rep.err
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
