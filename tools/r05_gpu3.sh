#!/bin/bash
# round 5, GPU call 3: the whole GPU suite, then the bench line
mkdir -p gpurun_out/r05c
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r05c/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05c/pytest_gpu.log
tail -5 gpurun_out/r05c/pytest_gpu.log
timeout 900 python bench.py --no-extras > gpurun_out/r05c/bench_line.json 2> gpurun_out/r05c/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05c/bench_line.json').read().strip().splitlines()[-1])
print('value', d['value'], d['dtype'], 'f32 draw', d.get('value_f32_draw'), 'clock', d.get('clock'), 'frac', d['roofline']['frac'], d['roofline'].get('frac_at_effective_clock'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
