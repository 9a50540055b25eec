#!/usr/bin/env python3
"""Do two Laplace contexts on two streams overlap on one MI355X when the table kernel leaves CUs free?  (round 6 feasibility probe for a
pipelined batch schedule: A(n+1) = conversion + pyrDowns beside B(n) = IIR / collapse / output.)  One host thread enqueues 32-frame calls
of two contexts alternately; aggregate frames/s against one context alone.  LVM_D0_FUSED_GROUPS = persistent groups of the table kernel."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.argv = [sys.argv[0]]
import bench  # noqa: E402

lvm = importlib.import_module("live-video-magnification_amd")


def run(nctx, K=12):
    Rs = [bench.Runner(lvm, torch, np, 1, None, 1, 32, 32, 0, [i], 32) for i in range(nctx)]
    for R in Rs:
        R.prime(K * 32, 64)
        R.run(64)
    torch.cuda.synchronize()
    for _ in range(2):
        for R in Rs:
            R.run(32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for R in Rs:
            R.run(32)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for R in Rs:
        R.close()
    return nctx * K * 32 / dt


if __name__ == "__main__":
    print("groups", os.environ.get("LVM_D0_FUSED_GROUPS", "all"), "one context %.0f fps, two contexts %.0f fps aggregate" % (run(1), run(2)))
