#!/bin/bash
# round 5, GPU call 7: the whole GPU suite after fastmc_create's generator default moved to the handle's precision; smoke; fuzz; bench line
mkdir -p gpurun_out/r05g
timeout 3000 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r05g/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05g/pytest_gpu.log
grep -n "FAILED\|passed\|failed\|pytest rc" gpurun_out/r05g/pytest_gpu.log | tail -35
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05g/smoke.log 2>&1; tail -1 gpurun_out/r05g/smoke.log
timeout 1500 python tools/fuzz_families.py 300 515 > gpurun_out/r05g/fuzz_300.txt 2>&1; tail -1 gpurun_out/r05g/fuzz_300.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r05g/bench_line.json 2> gpurun_out/r05g/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05g/bench_line.json').read().strip().splitlines()[-1])
print('value', d['value'], d['dtype'], 'f32 draw', d.get('value_f32_draw'), 'clock', d['clock']['effective_GHz'], 'frac', d['roofline']['frac'])
for k,v in d['extras'].items(): print(k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if not isinstance(vv,(dict,list,str)) or kk=='rows_kernel'})
PY
