#!/bin/sh
# TEST INFRASTRUCTURE: compiles the product sources against the HIP emulation header so the
# CPU test-suite can check kernel logic.  Output: tests/emu/_build/liblvm_emu.so (git-ignored).
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
src="$root/live-video-magnification_amd/csrc"
mkdir -p "$here/_build"
objs=""
for f in lvm_api.hip labconv.hip laplace.hip riesz.hip color.hip preprocess.hip compose.hip mjpeg.hip mjpeg_decode.hip lab_tables.cpp; do
  o="$here/_build/$f.o"
  if [ ! -f "$o" ] || [ "$src/$f" -nt "$o" ] || [ "$src/lvm_internal.h" -nt "$o" ] || [ "$src/pyramid.h" -nt "$o" ] || [ "$src/lab_lut.h" -nt "$o" ] || [ "$src/mjpeg_tables.h" -nt "$o" ] || [ "$here/include/hip/hip_runtime.h" -nt "$o" ] || [ "$here/include/lvm_gfx950.h" -nt "$o" ]; then
    g++ -x c++ -std=c++17 -O2 -march=x86-64-v3 -ffp-contract=off -fPIC -I"$here/include" -I"$root/include" -I"$src" \
        -Wno-unused-function -c "$src/$f" -o "$o" &
  fi
  objs="$objs $o"
done
wait
if [ ! -f "$here/_build/hip_emu.o" ] || [ "$here/hip_emu.cpp" -nt "$here/_build/hip_emu.o" ] || [ "$here/include/hip/hip_runtime.h" -nt "$here/_build/hip_emu.o" ]; then
  g++ -std=c++17 -O2 -fPIC -I"$here/include" -c "$here/hip_emu.cpp" -o "$here/_build/hip_emu.o"
fi
# relink only when an object changed: a loaded liblvm_emu.so must not be rewritten under another test process
so="$here/_build/liblvm_emu.so"
relink=0
[ -f "$so" ] || relink=1
for o in $objs "$here/_build/hip_emu.o"; do [ "$o" -nt "$so" ] && relink=1; done
if [ "$relink" = 1 ]; then
  g++ -shared -fPIC -Wl,-Bsymbolic -o "$so.tmp" $objs "$here/_build/hip_emu.o" && mv -f "$so.tmp" "$so"
fi
