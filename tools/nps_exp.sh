export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_npstream.py -x -q 2>&1 | tail -3
python3 tools/sameseed_rate.py 10000 100 2>&1 | grep -E "numpy|same"
python3 tools/sameseed_rate.py 40000 400 2>&1 | grep numpy
