# bench.py plain vs one-process-two-workers vs two ranks on ONE device (functional check of the N > 1 paths on a 1-GPU box)
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), d["n_gpus"], d["config"]["workers"], d["config"]["result_exchange"], d["config"]["histogram_total"])'
python bench.py --no-cpu-baseline --no-extras --no-sustained 2>/dev/null | grep "^{" | tail -1 | python -c "$P" plain
FASTMC_BENCH_DEVICES=0,0 python bench.py --gpus 2 --no-cpu-baseline --no-extras --no-sustained 2>/dev/null | grep "^{" | tail -1 | python -c "$P" threads
RANK=1 WORLD_SIZE=2 LOCAL_RANK=1 LOCAL_WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 FASTMC_BENCH_DEVICE=0 python bench.py --gpus 2 --no-cpu-baseline --no-extras --no-sustained > /dev/null 2>&1 &
RANK=0 WORLD_SIZE=2 LOCAL_RANK=0 LOCAL_WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 FASTMC_BENCH_DEVICE=0 python bench.py --gpus 2 --no-cpu-baseline --no-extras --no-sustained 2>/dev/null | grep "^{" | tail -1 | python -c "$P" ranks
wait
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --no-cpu-baseline --no-extras --no-sustained 2>/dev/null | grep "^{" | tail -1 | python -c "$P" torchrun1
