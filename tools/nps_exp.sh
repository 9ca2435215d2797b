RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
ls -la $RT
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verbosity=1
LD_PRELOAD=$RT python3 -c "print('hello under asan')"; echo "rc=$?"
export FASTMC_LIB=$PWD/build/asan/libfastmc_asan.so
LD_PRELOAD=$RT python3 -c "
import sys; sys.path.insert(0,'.')
from fast_amd import _lib
h=_lib.Handle(256,40,'f64',0)
print('handle ok', h.last_kernels())
"; echo "rc=$?"
