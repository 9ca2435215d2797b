#!/bin/bash
# round 5, GPU call 4: the whole GPU suite (all failures), then rocprofv3 profiles of the four priced row kernels
mkdir -p gpurun_out/r05d
timeout 3000 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/r05d/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05d/pytest_gpu.log
tail -25 gpurun_out/r05d/pytest_gpu.log
bash tools/profile_gpu.sh r05d > gpurun_out/r05d/prof_f64gen.log 2>&1
bash tools/profile_gpu.sh r05d_f32draw --rng-precision f32 > gpurun_out/r05d/prof_f32draw.log 2>&1
bash tools/profile_gpu.sh r05d_2048 --workload config3 > gpurun_out/r05d/prof_2048.log 2>&1
bash tools/profile_gpu.sh r05d_2048_f32draw --workload config3 --rng-precision f32 > gpurun_out/r05d/prof_2048_f32.log 2>&1
ls gpurun_out/prof_r05d*/
