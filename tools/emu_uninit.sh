#!/bin/sh
# TEST INFRASTRUCTURE: the CPU emulation build of the product sources (tests/emu) with every automatic variable pre-filled with a
# byte pattern (clang's -ftrivial-auto-var-init=pattern: floats become -1.7e38-like values, pointers non-canonical).  A host function or a
# kernel that reads a local before writing it computes garbage deterministically here, where the plain builds compute whatever the
# stack held -- usually something harmless.  That is how the uninitialised top knots of the inverse-gamma table were found in round 4
# (lvm_create's stack array, csrc/lab_tables.cpp spline_build): one GPU test run in four inside a full pytest session, never in isolation.
# Usage: tools/emu_uninit.sh [pytest args]; default = every emulation-based test (~7 min with -n 7).
set -e
root=$(cd "$(dirname "$0")/.." && pwd); here=$root/tests/emu; src=$root/live-video-magnification_amd/csrc
out=${LVM_UNINIT_DIR:-/tmp/lvm_emu_uninit}; mkdir -p "$out"
CXX=${LVM_CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}
for f in lvm_api.hip labconv.hip laplace.hip riesz.hip color.hip preprocess.hip compose.hip mjpeg.hip mjpeg_decode.hip lab_tables.cpp; do
  "$CXX" -x c++ -std=c++17 -O1 -march=x86-64-v3 -ffp-contract=off -fPIC -ftrivial-auto-var-init=pattern \
      -I"$here/include" -I"$root/include" -I"$src" -Wno-unused-function -c "$src/$f" -o "$out/$f.o" &
done
wait
"$CXX" -std=c++17 -O1 -fPIC -I"$here/include" -c "$here/hip_emu.cpp" -o "$out/hip_emu.o"
"$CXX" -shared -fPIC -Wl,-Bsymbolic -o "$out/liblvm_emu.so" "$out"/*.o
cd "$root"
[ $# -gt 0 ] || set -- tests/test_emu_bench_pattern.py tests/test_emu_parity.py tests/test_emu_random.py tests/test_compose.py tests/test_preprocess.py tests/test_export.py tests/test_host_display.py tests/test_mjpeg.py tests/test_mjpeg_decode.py -m "not gpu" -n 7
LVM_EMU_LIB="$out/liblvm_emu.so" python -m pytest -x -q "$@"
