#!/bin/bash
# A/B of row-kernel variants on ONE box, interleaved:  tools/ab_rows.sh "<lib> <lib> ..." [rounds]
# prints per library: it/s, rows ms and cols ms per step (10 000 iterations), configs[1]
LIBS=$1; R=${2:-3}
for i in $(seq $R); do
  for L in $LIBS; do
    FASTMC_LIB=$PWD/fast_amd/$L python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-sustained --no-f64-generator-pass --no-host-cost-pass 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), 'it/s  rows', round(d['pipeline']['rows_ms'],3), 'cols', round(d['pipeline']['cols_ms'],3), 'step', round(d['ms_per_step'],3))"
  done
done
