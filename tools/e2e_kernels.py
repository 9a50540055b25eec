#!/usr/bin/env python3
"""Per-kernel HIP-event times of lvm_process on page-locked frames (zero-copy aliases) -- where does a host -> host frame's time go?"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import bench
lvm = importlib.import_module("live-video-magnification_amd")
mode = sys.argv[1] if len(sys.argv) > 1 else "laplace"
ck, pk = lvm.synth.config(bench.MODES[mode]); clip = lvm.synth.Clip(**ck)
h, w = ck["h"], ck["w"]; fb = h * w * 3
lib = lvm.load()
pin, pout = C.c_void_p(), C.c_void_p()
lib.lvm_host_alloc(fb * 8, C.byref(pin)); lib.lvm_host_alloc(fb, C.byref(pout))
src = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(8, h, w, 3))
for t in range(8): src[t] = clip.frame(t)
cp = lvm.LvmParams(pk["mode"], pk["levels"], pk["amplification"], pk["coWavelength"], pk["coLow"], pk["coHigh"], pk["chromAttenuation"], pk["framerate"], 0)
ctx = lvm.Context(0, 1)
prod = C.c_int(0)
call = lambda i: lib.lvm_process(ctx.h, C.byref(cp), pin.value + (i % 8) * fb, w, h, 3, w * 3, pout.value, w * 3, C.byref(prod))
for i in range(40): call(i)
t0 = time.perf_counter()
for i in range(100): call(i)
dt = (time.perf_counter() - t0) / 100
ctx.profile(True)
for i in range(50): call(i)
prof = ctx.profile_collect(); ctx.profile(False)
print("%s: %.1f us per call; kernels (us per launch): %s ; sum %.1f" % (mode, dt * 1e6, " ".join("%s %.1f" % (k, 1e3 * ms / max(n, 1)) for k, (ms, n) in prof.items()),
      sum(1e3 * ms / max(n, 1) for ms, n in prof.values())))
