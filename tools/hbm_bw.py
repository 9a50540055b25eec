"""Practical HBM bandwidth of the box with PyTorch own copy / reduction / fill kernels (context for the roofline fractions:\nthe 8 TB/s peak of the data sheet is not what a device-to-device copy reaches).  Run on the GPU box: python tools/hbm_bw.py"""
import torch, time
d = torch.device("cuda:0")
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=d).normal_()
    b = torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("copy %5d MB: %.1f us, %.2f TB/s (read + write)" % (mb, ms * 1e3, 2 * n * 4 / (ms * 1e-3) / 1e12))
    e0.record()
    for _ in range(20): s = a.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("read %5d MB: %.1f us, %.2f TB/s" % (mb, ms * 1e3, n * 4 / (ms * 1e-3) / 1e12))
    e0.record()
    for _ in range(20): b.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("write %5d MB: %.1f us, %.2f TB/s" % (mb, ms * 1e3, n * 4 / (ms * 1e-3) / 1e12))
