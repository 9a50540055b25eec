#!/bin/sh
# TEST INFRASTRUCTURE: compiles the product sources against the HIP emulation header so the
# CPU test-suite can check kernel logic.  Output: tests/emu/_build/liblvm_emu.so (git-ignored).
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
src="$root/live-video-magnification_amd/csrc"
mkdir -p "$here/_build"
objs=""
for f in lvm_api.hip laplace.hip riesz.hip color.hip preprocess.hip compose.hip lab_tables.cpp; do
  o="$here/_build/$f.o"
  if [ ! -f "$o" ] || [ "$src/$f" -nt "$o" ] || [ "$src/lvm_internal.h" -nt "$o" ] || [ "$src/pyramid.h" -nt "$o" ] || [ "$here/include/hip/hip_runtime.h" -nt "$o" ]; then
    g++ -x c++ -std=c++17 -O2 -march=x86-64-v3 -ffp-contract=off -fPIC -I"$here/include" -I"$root/include" -I"$src" \
        -Wno-unused-function -c "$src/$f" -o "$o" &
  fi
  objs="$objs $o"
done
wait
g++ -std=c++17 -O2 -fPIC -I"$here/include" -c "$here/hip_emu.cpp" -o "$here/_build/hip_emu.o"
g++ -shared -fPIC -Wl,-Bsymbolic -o "$here/_build/liblvm_emu.so" $objs "$here/_build/hip_emu.o"
