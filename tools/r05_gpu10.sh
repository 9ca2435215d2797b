#!/bin/bash
# round 5, GPU call 10 (after the tile walk of rows and columns): profiles of the four priced row kernels, a bench line, 400 fuzz cases, soak
mkdir -p gpurun_out/r05r
bash tools/profile_gpu.sh r05r > gpurun_out/r05r/prof_f64gen.log 2>&1
bash tools/profile_gpu.sh r05r_f32draw --rng-precision f32 > gpurun_out/r05r/prof_f32draw.log 2>&1
bash tools/profile_gpu.sh r05r_2048 --workload config3 > gpurun_out/r05r/prof_2048.log 2>&1
bash tools/profile_gpu.sh r05r_2048_f32draw --workload config3 --rng-precision f32 > gpurun_out/r05r/prof_2048_f32.log 2>&1
timeout 900 python bench.py > gpurun_out/r05r/bench_line.json 2> gpurun_out/r05r/bench_line.err; python -c "
import json; d=json.loads(open('gpurun_out/r05r/bench_line.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['clock'], d['cpu_baseline']['value'], d['extras']['same_seed_1024'])"
timeout 1500 python tools/fuzz_families.py 400 20261005 > gpurun_out/r05r/fuzz_400.txt 2>&1; tail -3 gpurun_out/r05r/fuzz_400.txt
timeout 900 python tools/soak.py > gpurun_out/r05r/soak.txt 2>&1; tail -1 gpurun_out/r05r/soak.txt
