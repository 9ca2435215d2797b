export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_npstream.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed"
python3 tools/sameseed_rate.py 40000 400 2>&1 | grep numpy
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nps16 -- python3 tools/sameseed_rate.py 10000 100 > gpurun_out/nps16.txt 2>&1
cat gpurun_out/prof_nps16/*/*kernel_stats.csv | head -4 | cut -d, -f1-4
