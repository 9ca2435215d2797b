for k in 1 2; do
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['pipeline']['gpu_busy_ms_per_step'], d['ms_per_step'])"
FASTMC_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$k bench.py --gpus 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dist ', d['value'], d['pipeline']['gpu_busy_ms_per_step'], d['ms_per_step'], d['config']['result_exchange'])"
done
