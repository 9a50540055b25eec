#!/bin/sh
# TEST INFRASTRUCTURE: the CPU emulation build of the product sources (tests/emu) under AddressSanitizer.  In that build every
# "device" buffer is a host heap block and every kernel runs on the CPU, so ASan checks each global load / store of every kernel
# (and the host code of the C ABI) against the bounds of its allocation.  Stack instrumentation is off (the work-items are
# ucontext fibers on heap stacks).  Usage: tools/emu_asan.sh [pytest args]; default = every emulation-based test (~17 min).
set -e
root=$(cd "$(dirname "$0")/.." && pwd); here=$root/tests/emu; src=$root/live-video-magnification_amd/csrc
out=${LVM_ASAN_DIR:-/tmp/lvm_emu_asan}; mkdir -p "$out"
for f in lvm_api.hip labconv.hip laplace.hip riesz.hip color.hip preprocess.hip compose.hip mjpeg.hip mjpeg_decode.hip lab_tables.cpp; do
  g++ -x c++ -std=c++17 -O1 -g1 -march=x86-64-v3 -ffp-contract=off -fPIC -fsanitize=address --param asan-stack=0 \
      -I"$here/include" -I"$root/include" -I"$src" -Wno-unused-function -c "$src/$f" -o "$out/$f.o" &
done
wait
g++ -std=c++17 -O1 -g1 -fPIC -fsanitize=address --param asan-stack=0 -I"$here/include" -c "$here/hip_emu.cpp" -o "$out/hip_emu.o"
g++ -shared -fPIC -fsanitize=address -Wl,-Bsymbolic -o "$out/liblvm_emu.so" "$out"/*.o
cd "$root"
[ $# -gt 0 ] || set -- tests/test_emu_parity.py tests/test_emu_random.py tests/test_compose.py tests/test_preprocess.py tests/test_export.py tests/test_host_display.py tests/test_mjpeg.py tests/test_mjpeg_decode.py -m "not gpu"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LVM_EMU_LIB="$out/liblvm_emu.so" \
  python -m pytest -x -q "$@"
