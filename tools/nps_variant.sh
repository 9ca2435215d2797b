#!/bin/bash
# Builds a variant of libfastmc.so whose translation unit 0 (the numpy-stream kernels) is compiled with extra -D flags:
#   tools/nps_variant.sh NAME "-DNPS1_WAVES=4 -DNPS1_EXP_TIMES=2"   ->  build/variants/libfastmc_NAME.so   (use: FASTMC_LIB=build/variants/libfastmc_NAME.so)
# The other objects are the ones `make -C fast_amd/csrc all` left in fast_amd/csrc/obj.
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../fast_amd/csrc"
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 --offload-compress -ffp-contract=fast -fno-slp-vectorize -DFMC_SPLIT_BUILD -DFMC_TU=0 $FLAGS -c -o ../../build/variants/tu0_$NAME.o fastmc.hip
OBJS=$(for u in 1 2 3 4 5 6 7 8 9 10; do echo obj/fastmc_tu$u.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -fPIC -shared -o ../../build/variants/libfastmc_$NAME.so ../../build/variants/tu0_$NAME.o $OBJS -ldl
echo built build/variants/libfastmc_$NAME.so
