#!/bin/bash
# AddressSanitizer + UBSan over the kernel sources, on the CPU build (GPU sanitizers are not available on this pool):
# compiles csrc/*.hip for the fiber executor with -fsanitize=address,undefined into /tmp/mtn_asan/libmtn_emu.so and runs
# the kernel-level test files against it (every LDS / global access of every kernel variant the tests drive, with the
# dynamic shared memory of each launch ending at a poisoned byte: tests/emu/hipemu.h).
#   tools/asan_emu.sh [pytest args, default: the kernel test files]
set -e
cd "$(dirname "$0")/.."
O=/tmp/mtn_asan
mkdir -p $O
make -C hand-gesture-recognition_amd/csrc -j8 emu OBJDIR=$O/build OUT=$O \
  CXX="g++ -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined" > $O/build.log 2>&1
export MTN_EMU_LIB=$O/libmtn_emu.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1
ARGS=${@:-tests/test_conv_kernels.py tests/test_vit_kernels.py tests/test_bn_kernels.py tests/test_metrics.py tests/test_model_parity.py tests/test_row4_gelan.py tests/test_infer.py}
python -m pytest $ARGS -q -m "not gpu" -n 6 -p no:cacheprovider
