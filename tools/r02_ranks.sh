for r in 0 1; do
  RANK=$r WORLD_SIZE=2 LOCAL_RANK=$r LOCAL_WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 FASTMC_BENCH_DEVICE=0 FASTMC_RCCL_TIMEOUT=30 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-sustained > gpurun_out/ranks_$r.out 2> gpurun_out/ranks_$r.err &
done; wait
for r in 0 1; do echo "== rank $r"; grep -v "^Config" gpurun_out/ranks_$r.err | tail -15; tail -c 600 gpurun_out/ranks_$r.out; done
