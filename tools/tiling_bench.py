#!/usr/bin/env python3
"""What the spatial-tiling demonstrator costs on ONE GPU (no claim beyond that: a GPU per rank was never available): BASELINE configs[4]'s frame
(Riesz 3840 x 2160, 8 levels) per frame through the unsplit per-frame surface and through two stripes + gathered coarse levels in one process
(tiling.run_local's steps with device buffers; the exchanges are device-to-device row copies)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

lvm = importlib.import_module("live-video-magnification_amd")
T = lvm.tiling


def main():
    w, h, levels, n = 3840, 2160, 8, 24
    ck, pk = lvm.synth.config(2, (w, h, levels))
    clip = lvm.synth.Clip(**ck)
    frames = [clip.frame(t) for t in range(4)]
    mem = T._Torch(torch, 0)
    cp = lvm.LvmParams(pk["mode"], pk["levels"], pk["amplification"], pk["coWavelength"], pk["coLow"], pk["coHigh"], pk["chromAttenuation"], pk["framerate"], 0)
    # unsplit, device-resident
    ctx = lvm.Context(0, 1)
    d_in = [mem.from_host(f) for f in frames]
    d_out = mem.empty((h, w, 3), np.uint8)
    step = ctx.make_stepper(cp, w, h, 3, w * 3, w * h * 3, w * 3, w * h * 3, 0)
    for i in range(6):
        step(d_in[i % 4].data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(d_in[i % 4].data_ptr(), d_out.data_ptr())
    ctx.synchronize(); torch.cuda.synchronize()
    t_un = (time.perf_counter() - t0) / n
    ctx.close()
    # two stripes in one process
    for world in (2, 4):
        wk = [T.StripeWorker(lvm, r, world, w, h, pk, mem=mem) for r in range(world)]
        ext = [[mem.from_host(f[k.ext0:k.ext1]) for f in frames] for k in wk]

        def frame(i):
            for k, e in zip(wk, ext):
                k.stage1(e[i % 4])
            for k in wk:
                row0, rows = k.owned_residual_rows()
                mem.copy_rows(wk[0].octF, row0, rows, 0, rows.shape[0])
            wk[0].coarse_planes()
            for k, e in zip(wk, ext):
                mem.copy_rows(k.res_in, 0, wk[0].resF, k.f_ext0, k.rh)
                k.stage2(e[i % 4])
        for i in range(6):
            frame(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            frame(i)
        torch.cuda.synchronize()
        t_ti = (time.perf_counter() - t0) / n
        print("Riesz 3840x2160 L8, one GPU: unsplit per-frame call %.0f us; %d stripes + gathered coarse levels, every step synchronised, %.0f us per frame "
              "(stage 1 + gather + coarse + scatter + stage 2 of all stripes run one after another on the one device)" % (1e6 * t_un, world, 1e6 * t_ti))
        for k in wk:
            k.close()


if __name__ == "__main__":
    main()
