#!/bin/sh
# TEST INFRASTRUCTURE: the CPU emulation build of the product sources (tests/emu) with every __shared__ array in one linker section that the
# emulator fills with 0xFF (NaN as float) before EACH workgroup (-DHIPEMU_POISON_LDS): on the GPU a workgroup finds in LDS whatever the previous
# one -- of any kernel, of any process -- left there, in the plain emulation it finds its own previous values.  A kernel that reads LDS it has not
# written computes NaN / garbage here.  Clean on the 202 emulation tests at the end of round 4.  Single-threaded tests only (the arrays are not
# thread-local in this build).  Usage: tools/emu_lds_poison.sh [pytest args]
set -e
root=$(cd "$(dirname "$0")/.." && pwd); here=$root/tests/emu; src=$root/live-video-magnification_amd/csrc
out=${LVM_LDSP_DIR:-/tmp/lvm_emu_ldspoison}; mkdir -p "$out"
CXX=${LVM_CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}
for f in lvm_api.hip labconv.hip laplace.hip riesz.hip color.hip preprocess.hip compose.hip mjpeg.hip mjpeg_decode.hip lab_tables.cpp; do
  "$CXX" -x c++ -std=c++17 -O1 -march=x86-64-v3 -ffp-contract=off -fPIC -DHIPEMU_POISON_LDS=1 \
      -I"$here/include" -I"$root/include" -I"$src" -Wno-unused-function -c "$src/$f" -o "$out/$f.o" &
done
wait
"$CXX" -std=c++17 -O1 -fPIC -DHIPEMU_POISON_LDS=1 -I"$here/include" -c "$here/hip_emu.cpp" -o "$out/hip_emu.o"
"$CXX" -shared -fPIC -Wl,-Bsymbolic -o "$out/liblvm_emu.so" "$out"/*.o
cd "$root"
[ $# -gt 0 ] || set -- tests/test_emu_bench_pattern.py tests/test_emu_parity.py tests/test_compose.py tests/test_preprocess.py tests/test_export.py tests/test_mjpeg.py tests/test_mjpeg_decode.py -m "not gpu" -n 7
LVM_EMU_LIB="$out/liblvm_emu.so" python -m pytest -x -q "$@"
