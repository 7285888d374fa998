#!/bin/bash
# CPU sanitizer recipe ("-fsanitize=address host build").  Builds the host emulator of the kernel
# bodies (tests/emu/emu.cpp: the same templates the HIP kernels are made of, one OS thread per GPU thread) and the
# MINPACK restatement (csrc/gauss_fit.cpp) with AddressSanitizer + UndefinedBehaviorSanitizer and runs the tests that
# drive them -- every index computation of the convolution kernels, the Poisson sampler and the Gaussian fit -- under
# the sanitizer runtime.  CPU build only: never on the GPU box (gpurun refuses GPU sanitizer runs).
#
#     tools/asan_emu.sh [extra pytest arguments]        (~8 min to compile at -O1, a few minutes to run)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/build/asan"
mkdir -p "$OUT"
FLAGS="-O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unknown-pragmas -pthread -fsanitize=address,undefined -fno-omit-frame-pointer"
echo "building $OUT/libemu.so (sanitized)"
g++ $FLAGS "$ROOT/tests/emu/emu.cpp" -o "$OUT/libemu.so"
echo "building $OUT/libgaussfit.so (sanitized)"
g++ $FLAGS -I"$ROOT/include" "$ROOT/rescan_line_sted_amd/csrc/gauss_fit.cpp" "$ROOT/tools/asan_gauss_fit_main.cpp" -o "$OUT/libgaussfit.so"
ASAN_LIB="$(g++ -print-file-name=libasan.so)"
UBSAN_LIB="$(g++ -print-file-name=libubsan.so)"
export LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB"
# python itself leaks by design; numpy allocates before the runtime is up: report real errors only
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
export RLSTED_EMU_LIB="$OUT/libemu.so"
export RLSTED_GAUSSFIT_LIB="$OUT/libgaussfit.so"
cd "$ROOT"
python -m pytest tests/test_emulated_kernels.py tests/test_poisson_spec.py tests/test_asan_gauss_fit.py -x -q -p no:cacheprovider "$@"
echo "sanitizer run clean"
