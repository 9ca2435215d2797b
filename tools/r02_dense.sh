#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=${1:-fast_amd/libfastmc.so}
for rep in 1 2 3; do
  FASTMC_NO_DENSE16=1 bash tools/abl.sh $L 2>&1 | grep rows | sed 's/^/12-wave /'
  FASTMC_NO_DENSE16=0 bash tools/abl.sh $L 2>&1 | grep rows | sed 's/^/16-wave /'
done
FASTMC_LIB=$PWD/$L python -m pytest tests/test_gpu_parity.py -x -q -k "test_device_rng_run_matches_oracle or test_device_rng_invariant or test_full_size_properties or test_fast_device_mode or statistics_match_host" 2>&1 | grep -E "passed|failed"
