#!/usr/bin/env python3
"""Generates tests/golden/ref_slices.json from the reference's OWN code.

Runs in the build container only (needs /root/reference): oracle/Makefile compiles the
OpenCV-free slices of the reference in place (TemporalFilter.cpp:82-297 butterworth +
getOptimalBufferSize, MagnificationParamsUi.hpp:27-34 motionHzToBlend) into
oracle/_ref/libref_slices.so; this script calls them and stores the answers.
The rest of the reference hot path needs OpenCV 4 and cannot be executed here.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

R = po.ref_slices()
assert R is not None, "build oracle/_ref first (make -C oracle ref)"
out = {"butterworth2": [], "optimal_buffer_size": {}, "motion_hz_to_blend": []}
wns = [0.5 / 15, 10 / 15, 1 / 15, 5 / 15, 0.5 / 30, 10 / 30, 0.84 / 15, 1.43 / 15, 0.001, 0.25, 0.5, 0.75, 0.999,
       0.0, 1.0, 1.5, -0.1]
for wn in wns:
    a, b = np.zeros(3), np.zeros(3)
    R.ref_butterworth(2, wn, a, b)
    out["butterworth2"].append({"Wn": wn, "a": [repr(float(x)) for x in a], "b": [repr(float(x)) for x in b]})
for fps in list(range(0, 130)) + [240, 300, 1000]:
    out["optimal_buffer_size"][str(fps)] = int(R.ref_getOptimalBufferSize(fps))
for hz, fps in [(0.4, 30), (3, 30), (1, 30), (5, 30), (0.0, 30), (-1, 30), (1, 0), (100, 30), (0.4, 60), (3, 60), (14.9, 30)]:
    out["motion_hz_to_blend"].append({"hz": hz, "fps": fps, "blend": repr(float(R.ref_motionHzToBlend(hz, fps)))})
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_slices.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote ref_slices.json")
