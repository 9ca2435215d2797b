#!/bin/bash
# Interleaved timing of several builds on ONE box: tools/abn.sh <rounds> lib1.so lib2.so ... [-- bench args]
R=$1; shift; LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for i in $(seq $R); do for L in "${LIBS[@]}"; do
  FASTMC_LIB=$PWD/$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), 'it/s  rows', round(d['pipeline']['rows_ms'],3), 'cols', round(d['pipeline']['cols_ms'],3), 'step', round(d['ms_per_step'],3))"
done; done
