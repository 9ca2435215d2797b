export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_npstream.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for v in 0 1; do
  export FASTMC_NPS_TWO_STREAMS=$v
  echo "== two streams: $v"
  python3 tools/sameseed_rate.py 40000 400 2>&1 | grep numpy
done
done
