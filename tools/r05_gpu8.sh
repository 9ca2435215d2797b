#!/bin/bash
# round 5, GPU call 8: statistical battery on the float64 generator's own draws, soak of 150 objects, the GPU suite on the UBSan host build
mkdir -p gpurun_out/r05h
timeout 900 python tools/generator_quality.py 32 1024 20261004 > gpurun_out/r05h/generator_quality_f64.txt 2>&1; tail -4 gpurun_out/r05h/generator_quality_f64.txt
timeout 900 python tools/generator_quality.py 32 1024 777 > gpurun_out/r05h/generator_quality_f64_seed777.txt 2>&1; tail -2 gpurun_out/r05h/generator_quality_f64_seed777.txt
timeout 900 python tools/soak.py > gpurun_out/r05h/soak.txt 2>&1; tail -1 gpurun_out/r05h/soak.txt
bash tools/ubsan_host.sh tests -m gpu -q -x > gpurun_out/r05h/ubsan_suite.txt 2> gpurun_out/r05h/ubsan_stderr.txt; tail -3 gpurun_out/r05h/ubsan_suite.txt; grep -c "runtime error" gpurun_out/r05h/ubsan_stderr.txt gpurun_out/r05h/ubsan_suite.txt
