#!/bin/sh
# TEST INFRASTRUCTURE: the CPU emulation build of the product sources (tests/emu) under clang's UndefinedBehaviorSanitizer: signed overflow,
# shifts, misaligned or out-of-bounds accesses of host code and kernels abort the test.  (pointer-overflow is off: kernels form addresses
# from null pointers of planes a variant never reads.)  Clean on the 202 emulation tests at the end of round 4.
# Usage: tools/emu_ubsan.sh [pytest args]; default = the emulation-based tests (~6 min with -n 7).
set -e
root=$(cd "$(dirname "$0")/.." && pwd); here=$root/tests/emu; src=$root/live-video-magnification_amd/csrc
out=${LVM_UBSAN_DIR:-/tmp/lvm_emu_ubsan}; mkdir -p "$out"
CXX=${LVM_CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}
for f in lvm_api.hip labconv.hip laplace.hip riesz.hip color.hip preprocess.hip compose.hip mjpeg.hip mjpeg_decode.hip lab_tables.cpp; do
  "$CXX" -x c++ -std=c++17 -O1 -march=x86-64-v3 -ffp-contract=off -fPIC -g -fsanitize=undefined -fno-sanitize=vptr,pointer-overflow -fno-sanitize-recover=undefined \
      -I"$here/include" -I"$root/include" -I"$src" -Wno-unused-function -c "$src/$f" -o "$out/$f.o" &
done
wait
"$CXX" -std=c++17 -O1 -fPIC -fsanitize=undefined -fno-sanitize=vptr,pointer-overflow -I"$here/include" -c "$here/hip_emu.cpp" -o "$out/hip_emu.o"
"$CXX" -shared -fPIC -fsanitize=undefined -fno-sanitize=vptr,pointer-overflow -Wl,-Bsymbolic -o "$out/liblvm_emu.so" "$out"/*.o
cd "$root"
[ $# -gt 0 ] || set -- tests/test_emu_bench_pattern.py tests/test_emu_parity.py tests/test_compose.py tests/test_preprocess.py tests/test_export.py tests/test_mjpeg.py tests/test_mjpeg_decode.py -m "not gpu" -n 7
rt=$(ls "$(dirname "$CXX")"/../lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
LD_PRELOAD="$rt" UBSAN_OPTIONS=print_stacktrace=1 LVM_EMU_LIB="$out/liblvm_emu.so" python -m pytest -x -q "$@"
