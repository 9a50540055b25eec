#!/usr/bin/env python3
"""Per-kernel HIP-event times of the per-frame schedule (T = 1, one stream) of a mode: python tools/t1_kernels.py MODE [W H LEVELS]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
args = sys.argv[1:]
sys.argv = [sys.argv[0]]
import bench
lvm = importlib.import_module("live-video-magnification_amd")
mode = args[0]
small = tuple(int(x) for x in args[1:4]) if len(args) >= 4 else None
cfg = {"laplace": 1, "riesz": 2, "color": 3}[mode]
R = bench.Runner(lvm, torch, np, cfg, small, 1, 32, 1, 0, [0], 32)
bench.timed_run(lvm, torch, R, 200, 32 if mode != "color" else 200, None, None)
import time
torch.cuda.synchronize(); t0 = time.perf_counter(); R.run(200); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
R.ctx.profile(True); R.run(40); torch.cuda.synchronize(); prof = R.ctx.profile_collect(); R.ctx.profile(False)
print(mode, small, "%.1f us per frame back to back; bracketed per kernel (us):" % (1e6 * dt), {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in prof.items()}, "launches per frame %.1f" % (sum(v[1] for v in prof.values()) / 40.0))
