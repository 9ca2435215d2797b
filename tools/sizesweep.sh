#!/bin/bash
# f64 iterations/s over grid sizes on one box: tools/sizesweep.sh [sizes...]
for n in "${@:-128 164 192 256 320 384 448 512 576 640 768 896 1000 1024 1152 1280 1536 1792 2048}"; do for s in $n; do
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --npxls $s 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($s, d['config']['kernel_path'], round(d['value']), 'it/s  rows', round(d['pipeline']['rows_ms'],3), 'cols', round(d['pipeline']['cols_ms'],3))"
done; done
