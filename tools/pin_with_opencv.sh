#!/bin/bash
# Pins the oracle at the OpenCV boundary.  Run ONCE on any box that has an OpenCV 4 development install, a C++20 g++ and the
# reference checkout (REF=/path/to/Live-Video-Magnification, default /root/reference):
#     bash tools/pin_with_opencv.sh
#   1. builds oracle/_ref/libref_magnify.so = the reference's own MagnificationProcessor.cpp + magnification/*.cpp compiled WHERE THEY LIE
#      + oracle/ref_driver.cpp, linked with this box's OpenCV (oracle/Makefile target ref_full);
#   2. recovers the build's forward Lab table and renders BASELINE configs 0-3 with the real reference (tools/pin_generate.py)
#      -> tests/golden/frames_cfg{0,1,2,3}.npz (a few MB: commit them);
#   3. runs the pin tests: oracle AND library (emulation build; the gfx950 library where a GPU exists) against those files.
# From then on `pytest -m refpin` is a parity gate that needs no OpenCV.
set -e
cd "$(dirname "$0")/.."
REF=${REF:-/root/reference}
make -C oracle REF="$REF" liblvm_oracle.so ref_full
test -f oracle/_ref/libref_magnify.so || { echo "no OpenCV 4 found (pkg-config opencv4 / /usr/include/opencv4): nothing pinned"; exit 2; }
python3 tools/pin_generate.py
python3 -m pytest tests/test_refpin.py -m refpin -q -s
