#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
O=gpurun_out/r02_abl2.txt
: > $O
for rep in 1 2; do
  bash tools/abl.sh fast_amd/libfastmc_r01.so fast_amd/libfastmc_base.so fast_amd/libfastmc_b2.so fast_amd/libfastmc_b3.so fast_amd/libfastmc_nobitop.so fast_amd/libfastmc_nophilox.so fast_amd/libfastmc_seed7.so 2>&1 | grep rows >> $O
done
FASTMC_LIB=$PWD/fast_amd/libfastmc_b3.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "test_device_generator_matches_oracle_restatement or test_device_rng_run_matches_oracle or test_device_generator_statistical_quality or test_device_rng_invariant or test_full_size_properties" >> $O 2>&1
cat $O
