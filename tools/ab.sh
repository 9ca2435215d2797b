#!/bin/bash
# A/B timing on ONE box, interleaved:  tools/ab.sh <libA.so> <libB.so> [rounds] [bench args]
A=$1; B=$2; R=${3:-3}; shift 3 || true
for i in $(seq $R); do
  for L in $A $B; do
    FASTMC_LIB=$PWD/$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), 'it/s  rows', round(d['pipeline']['rows_ms'],3), 'cols', round(d['pipeline']['cols_ms'],3), 'step', round(d['ms_per_step'],3))"
  done
done
