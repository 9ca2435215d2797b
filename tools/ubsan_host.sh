#!/bin/bash
# The library's HOST code (handles, queues, dispatch, communicator glue: 2.4 k lines of C++ in fastmc.hip) under UBSan on the GPU box;
# the device code is the normal gfx950 code (-fno-gpu-sanitize).  (AddressSanitizer is not an option here: ROCm's ASan runtime
# intercepts hsa_amd_memory_pool_allocate and needs XNACK, which this pool does not offer -- tried, the first hipMalloc fails.)
#   build (here or on the box, ~2 min):   tools/ubsan_host.sh build
#   run   (GPU box):                      tools/ubsan_host.sh tests/test_gpu_parity_generator.py -q     -> reports go to stderr, run continues
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-}" == "build" ]; then
  mkdir -p $ROOT/build/ubsan && cd $ROOT/fast_amd/csrc
  for u in 0 1 2 3 4 5 6 7 8 9 10; do echo $u; done | xargs -P 8 -I{} /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 --offload-compress \
    -ffp-contract=fast -fno-slp-vectorize -fsanitize=undefined,bounds,float-divide-by-zero -fno-gpu-sanitize -fno-omit-frame-pointer -DFMC_SPLIT_BUILD -DFMC_TU={} \
    -c -o ../../build/ubsan/tu{}.o fastmc.hip
  # (a shared object does not get the runtime by itself: the static archives go in whole)
  A=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.ubsan_standalone-x86_64.a); B=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.ubsan_standalone_cxx-x86_64.a)
  cd $ROOT/build/ubsan && /opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -fPIC -shared -fno-gpu-sanitize -o libfastmc_ubsan.so tu*.o \
    -Wl,--whole-archive $A $B -Wl,--no-whole-archive -ldl -lpthread
  exit $?
fi
export FASTMC_LIB=$ROOT/build/ubsan/libfastmc_ubsan.so
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
cd $ROOT && timeout 2400 python -m pytest "$@"
