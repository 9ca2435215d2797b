#!/bin/bash
# iterations/s over grid sizes on one box, both generator precisions: tools/sizesweep.sh [sizes...]
# (eight timed steps after three warm-up steps -- with one warm-up step the first touches of a fresh V slab sit in the timed steps and small grids read 5-9 % low; 10 000 iterations per step, Np = 82, float64 pipeline; per size: it/s, it/s x N^2 relative to the 1024^2 figure comes from the table)
for n in "${@:-128 192 256 320 384 448 512 576 640 768 896 1000 1024 1152 1280 1536 1792 2000 2048}"; do for s in $n; do for prec in f64 f32; do
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-sustained --no-f32-draw-pass --no-host-cost-pass --rng-precision $prec --npxls $s 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print($s, '$prec-gen', round(d['value']), 'it/s', 'x N^2 = %.3g' % (d['value'] * $s * $s), d['roofline']['kernel'], 'rows', round(d['pipeline']['rows_ms'],2), 'cols', round(d['pipeline']['cols_ms'],2))"
done; done; done
