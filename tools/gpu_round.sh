#!/bin/bash
# ONE parametrised GPU batch (runs ON THE GPU BOX via gpurun) instead of a script per call:
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh <tag> <step> [<step> ...]'
# Everything a step writes lands under gpurun_out/<tag>/ (merged back by gpurun); copy what is to be judged into profiles/.
# Steps (run in the order given):
#   suite                 python -m pytest tests -m gpu -q --maxfail=30
#   smoke                 __graft_entry__.smoke()
#   bench | bench-nocpu   the full bench line (extras, CPU baseline) | without the CPU baseline
#   bench:<args>          bench.py with these arguments (commas for spaces), e.g. bench:--workload,config3,--no-extras
#   fuzz:<n>:<seed>       tools/fuzz_families.py n seed
#   soak                  tools/soak.py
#   quality:<seed>        tools/generator_quality.py 32 1024 seed
#   profile[:<args>]      tools/profile_gpu.sh <tag>[_<args>] <args>  (commas for spaces): kernel trace + PMC passes of the bench step
#   nps-profile           tools/profile_numpy_stream.sh <tag>_nps
#   sweep[:<sizes>]       tools/sizesweep.sh (comma-separated sizes; default list)
#   trace:<sizes>         tools/trace_sizes.sh <tag> sizes (per-kernel times at these grid sizes)
#   ubsan                 tools/ubsan_host.sh tests -m gpu -q (all but the rate assertions of the bench-contract file: that build is -O1)
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  case $name in
    suite) timeout 3000 python -m pytest tests -m gpu -q --maxfail=30 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
           grep -n "FAILED\|passed\|failed\|pytest rc" $OUT/pytest_gpu.log | tail -35 ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log ;;
    bench|bench-nocpu)
           extra=${arg//,/ }; [ $name == bench-nocpu ] && extra="--no-cpu-baseline $extra"
           timeout 1500 python bench.py $extra > $OUT/bench_line.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.err
           python - $OUT/bench_line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', round(d['value']), d['dtype'], 'ms/step', round(d['ms_per_step'], 3), 'f32 draw', d.get('value_f32_draw'), 'clock', d.get('clock', {}).get('effective_GHz'),
      'frac', d['roofline']['frac'], d['roofline'].get('frac_at_effective_clock'), 'cpu', d.get('cpu_baseline', {}).get('value'))
for k, v in d.get('extras', {}).items():
    print(' ', k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if not isinstance(vv, (dict, list, str))})
PY
           ;;
    fuzz)  n=${arg%%:*}; seed=${arg#*:}; timeout 2400 python tools/fuzz_families.py $n $seed > $OUT/fuzz_${n}_cases.txt 2>&1; tail -2 $OUT/fuzz_${n}_cases.txt; grep -c BAD $OUT/fuzz_${n}_cases.txt ;;
    soak)  timeout 900 python tools/soak.py > $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt ;;
    quality) timeout 900 python tools/generator_quality.py 32 1024 $arg > $OUT/generator_quality_seed$arg.txt 2>&1; tail -3 $OUT/generator_quality_seed$arg.txt ;;
    profile) a=${arg//,/ }; t=$TAG; [ -n "$arg" ] && t=${TAG}_$(echo "$arg" | tr -cd 'a-z0-9'); bash tools/profile_gpu.sh $t $a > $OUT/profile_$t.log 2>&1; grep "k_rows\|k_cols" gpurun_out/prof_$t/summary.md | head -12 ;;
    nps-profile) bash tools/profile_numpy_stream.sh ${TAG}_nps > $OUT/profile_nps.log 2>&1; tail -5 $OUT/profile_nps.log ;;
    sweep) bash tools/sizesweep.sh ${arg//,/ } > $OUT/sizesweep.txt 2>&1; cat $OUT/sizesweep.txt ;;
    trace) bash tools/trace_sizes.sh $TAG ${arg//,/ } > $OUT/trace_sizes.txt 2>&1; cat $OUT/trace_sizes.txt ;;
    ubsan) [ -f build/ubsan/libfastmc_ubsan.so ] || bash tools/ubsan_host.sh build > $OUT/ubsan_build.log 2>&1   # (build/ does not travel: built on the box, ~2 min)
           bash tools/ubsan_host.sh tests -m gpu -q --deselect tests/test_gpu_bench_contract.py > $OUT/ubsan_suite.txt 2> $OUT/ubsan_stderr.txt; tail -3 $OUT/ubsan_suite.txt; grep -c "runtime error" $OUT/ubsan_stderr.txt $OUT/ubsan_suite.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
