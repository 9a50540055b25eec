#!/usr/bin/env python3
"""A/B of the host -> host surfaces on a GPU box: lvm_process on page-locked / pageable 1080p frames and lvm_export_frames on
32-frame batches, under the environment given on the command line (LVM_ZERO_COPY, LVM_SPIN_WAIT, LVM_EXPORT_CHUNK).
    python tools/host_surfaces_ab.py [mode]      prints one line"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

lvm = importlib.import_module("live-video-magnification_amd")
mode = sys.argv[1] if len(sys.argv) > 1 else "laplace"
cfg = bench.MODES[mode]
ck, pk = lvm.synth.config(cfg)
clip = lvm.synth.Clip(**ck)
host = np.stack([clip.frame(t) for t in range(8)])
cp = lvm.LvmParams(pk["mode"], pk["levels"], pk["amplification"], pk["coWavelength"], pk["coLow"], pk["coHigh"], pk["chromAttenuation"], pk["framerate"], 0)
ctx = lvm.Context(0, 1)
twin = lvm.load().lvm_optimal_buffer_size(int(pk["framerate"]))
e = bench.e2e_host_record(lvm, np, ctx, C.byref(cp), host, Ke=200, warm=(twin + 12 if mode == "color" else 30))
ctx.close()
x = bench.export_host_record(lvm, np, 0, C.byref(cp), host, Kx=8)
env = " ".join("%s=%s" % (k, os.environ[k]) for k in ("LVM_ZERO_COPY", "LVM_SPIN_WAIT", "LVM_EXPORT_CHUNK") if k in os.environ)
print("%-8s %-44s e2e pageable %7.1f us  pinned %7.1f us (%5.0f fps) | export %7.1f us/frame %6.0f fps %5.1f GB/s" % (
    mode, env or "(defaults)", e["pageable"]["us_per_frame"], e["pinned"]["us_per_frame"], e["pinned"]["value"], x.get("us_per_frame", -1), x.get("value", -1), x.get("pcie_gbs", -1)))
