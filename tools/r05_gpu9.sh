#!/bin/bash
# round 5, GPU call 9 (after the tile walk of k_rows_wave): the whole GPU suite (all failures), then rocprofv3 profiles of the four priced row kernels
mkdir -p gpurun_out/r05n
timeout 3000 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/r05n/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05n/pytest_gpu.log
tail -25 gpurun_out/r05n/pytest_gpu.log
bash tools/profile_gpu.sh r05n > gpurun_out/r05n/prof_f64gen.log 2>&1
bash tools/profile_gpu.sh r05n_f32draw --rng-precision f32 > gpurun_out/r05n/prof_f32draw.log 2>&1
bash tools/profile_gpu.sh r05n_2048 --workload config3 > gpurun_out/r05n/prof_2048.log 2>&1
bash tools/profile_gpu.sh r05n_2048_f32draw --workload config3 --rng-precision f32 > gpurun_out/r05n/prof_2048_f32.log 2>&1
ls gpurun_out/prof_r05n*/
