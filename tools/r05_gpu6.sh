#!/bin/bash
# round 5, GPU call 6: the whole GPU suite, a 600-case differential fuzz of the kernel families (float32 draw and the fused float64
# generator against the direct family), the full bench line with its extras
mkdir -p gpurun_out/r05f
timeout 3000 python -m pytest tests -m gpu -q --maxfail=20 > gpurun_out/r05f/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05f/pytest_gpu.log
tail -6 gpurun_out/r05f/pytest_gpu.log
timeout 2400 python tools/fuzz_families.py 600 20261004 > gpurun_out/r05f/fuzz_600.txt 2>&1; tail -2 gpurun_out/r05f/fuzz_600.txt; grep -c BAD gpurun_out/r05f/fuzz_600.txt
timeout 1500 python bench.py > gpurun_out/r05f/bench_line.json 2> gpurun_out/r05f/bench.err; tail -c 600 gpurun_out/r05f/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f/bench_line.json').read().strip().splitlines()[-1])
print('value', d['value'], d['dtype'], 'f32 draw', d.get('value_f32_draw'), 'clock', d['clock']['effective_GHz'], 'frac', d['roofline']['frac'], d['roofline'].get('frac_at_effective_clock'), 'traffic', d['roofline']['traffic'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores',{}).get('value'), 'speedup', d['speedup_vs_cpu_1core'])
for k,v in d['extras'].items(): print(k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if not isinstance(vv,(dict,list,str)) or kk=='rows_kernel'})
PY
