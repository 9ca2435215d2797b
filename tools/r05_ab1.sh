#!/bin/bash
# round 5, GPU call 1: smoke, float64-generator parity tests, A/B of the round-4 library against the new generator
set -x
mkdir -p gpurun_out/r05a
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05a/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r05a/smoke.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "float64 or fused or generator or rng or smoke" > gpurun_out/r05a/pytest_gen.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/pytest_gen.log
for rep in 1 2; do
  FASTMC_LIB=$PWD/fast_amd/libfastmc_r04.so timeout 600 python bench.py --no-extras --no-cpu-baseline --no-sustained --steps 10 > gpurun_out/r05a/bench_r04lib_$rep.json 2> gpurun_out/r05a/bench_r04lib_$rep.err
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-sustained --steps 10 > gpurun_out/r05a/bench_new_$rep.json 2> gpurun_out/r05a/bench_new_$rep.err
done
tail -3 gpurun_out/r05a/smoke.log gpurun_out/r05a/pytest_gen.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'value', round(d['value']), d['dtype'], 'rows_ms', d['pipeline']['rows_ms'], 'frac', d['roofline']['frac'], 'f32', d.get('value_f32_draw'), d['roofline']['kernel'])
    except Exception as e:
        print(f, 'ERR', e)
PY
