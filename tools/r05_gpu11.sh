#!/bin/bash
# round 5, GPU call 11: the four priced kernels profiled on the LAST library of the round (tile walk, non-temporal V / coefficient loads)
mkdir -p gpurun_out/r05final
bash tools/profile_gpu.sh r05final > gpurun_out/r05final/prof_f64gen.log 2>&1
bash tools/profile_gpu.sh r05final_f32draw --rng-precision f32 > gpurun_out/r05final/prof_f32draw.log 2>&1
bash tools/profile_gpu.sh r05final_2048 --workload config3 > gpurun_out/r05final/prof_2048.log 2>&1
bash tools/profile_gpu.sh r05final_2048_f32draw --workload config3 --rng-precision f32 > gpurun_out/r05final/prof_2048_f32.log 2>&1
bash tools/profile_numpy_stream.sh r05final_nps > /dev/null 2>&1
grep "k_rows_wave\|k_cols_wave\|onepass" gpurun_out/prof_r05final*/summary.md | head -20
