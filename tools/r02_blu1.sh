#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in "$@"; do
for n in ${SIZES:-1000 1200 1500 1900}; do
FASTMC_LIB=$PWD/$L python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-sustained --npxls $n 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', $n, d['config']['kernel_path'], round(d['value']), 'it/s  rows', round(d['pipeline']['rows_ms'],3), 'cols', round(d['pipeline']['cols_ms'],3))"
done; done
