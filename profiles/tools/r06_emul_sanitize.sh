#!/bin/bash
# Round 6: is there undefined behaviour in the lane program's SOURCE?  The CPU wavefront emulator (tests/emul: the same gn_lane.h / gn_woodbury.h /
# gn_backward.h / gn_long.h the kernels are compiled from) built with clang's AddressSanitizer + UndefinedBehaviorSanitizer, and a second time with every
# uninitialised automatic variable poisoned (-ftrivial-auto-var-init=pattern: floating-point locals start as NaN, integers / pointers as 0xAA..), then driven
# through the emulator test-suite: an out-of-bounds index, a read of an unwritten register array or lane slot shows up as a sanitizer report or a changed result.
cd "$(dirname "$0")/../.."
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname $($CLANG -print-file-name=libclang_rt.asan-x86_64.so))
mode=${1:-asan}; shift
if [ "$mode" = asan ]; then
  export DGP_EMUL_CXX=$CLANG DGP_EMUL_LIB=$PWD/tests/emul/libgn_emul_asan.so
  export DGP_EMUL_FLAGS="-fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libasan -fno-omit-frame-pointer -g"
  export LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0 UBSAN_OPTIONS=print_stacktrace=1
else
  export DGP_EMUL_CXX=$CLANG DGP_EMUL_LIB=$PWD/tests/emul/libgn_emul_poison.so
  export DGP_EMUL_FLAGS="-ftrivial-auto-var-init=pattern"
fi
python -m pytest tests/test_lane_emulator.py -x -q "$@"
