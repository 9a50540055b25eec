#!/usr/bin/env python3
"""Motion-JPEG encode on the device, measured (run on a GPU box: `gpurun -- 'python tools/mjpeg_bench.py > gpurun_out/mjpeg.txt'`):
  1. lvm_mjpeg_encode_device on device-resident 3840 x 1080 canvases (the export's side-by-side shape): frames/s, bytes per frame, the
     five kernels (HIP events, lvm_profile_*);
  2. the export surface host -> host: lvm_export_frames (canvases down, 12.4 MB each) against lvm_export_frames_mjpeg (JPEG frames down),
     page-locked buffers, 32 frames per call;
  3. libjpeg (Pillow) encoding the same canvas on one host core, for scale.
"""
import ctypes as C
import importlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lvm = importlib.import_module("live-video-magnification_amd")


def main():
    import torch
    lib = lvm.load()
    q = int(os.environ.get("MJ_QUALITY", "85"))
    ck, pk = lvm.synth.config(1)                      # 1080p Laplace clip
    clip = lvm.synth.Clip(seed=1234, **ck)
    w, h = ck["w"], ck["h"]
    frames = np.stack([clip.frame(t) for t in range(8)])
    canvas = np.concatenate([frames, frames[::-1]], axis=2)          # 3840 x 1080 side by side
    cw, chh = 2 * w, h
    T = 8
    d = torch.from_numpy(canvas).cuda()
    ctx = lvm.Context(0, 1)
    out = ctx.mjpeg_encode_device(C.c_void_p(d.data_ptr()), cw, chh, T, quality=q)
    sizes = [len(x) for x in out]
    print("canvas %dx%d quality %d: %.0f KB per frame (%.2f bits/pixel, %.1f x smaller than the 12.4 MB canvas)" % (
        cw, chh, q, np.mean(sizes) / 1e3, 8 * np.mean(sizes) / (cw * chh), cw * chh * 3 / np.mean(sizes)))
    cap = int(lib.lvm_mjpeg_bound(cw, chh)) * T
    buf = np.empty(cap, np.uint8)
    offs = (C.c_size_t * (T + 1))()
    for _ in range(3):
        ctx._check(lib.lvm_mjpeg_encode_device(ctx.h, d.data_ptr(), cw, chh, cw * 3, cw * 3 * chh, T, q, buf.ctypes.data, cap, offs))
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        ctx._check(lib.lvm_mjpeg_encode_device(ctx.h, d.data_ptr(), cw, chh, cw * 3, cw * 3 * chh, T, q, buf.ctypes.data, cap, offs))
    dt = time.perf_counter() - t0
    print("lvm_mjpeg_encode_device (device frames in, JPEG on the host out, %d frames per call, pageable output): %.0f frames/s, %.1f us per frame" % (
        T, K * T / dt, 1e6 * dt / (K * T)))
    ctx.profile(True)
    for _ in range(5):
        ctx._check(lib.lvm_mjpeg_encode_device(ctx.h, d.data_ptr(), cw, chh, cw * 3, cw * 3 * chh, T, q, buf.ctypes.data, cap, offs))
    prof = ctx.profile_collect()
    ctx.profile(False)
    tot = 0.0
    for name, (ms, cnt) in prof.items():
        if name.startswith("mj_"):
            print("  %-14s %8.1f us per launch of %d frames (%d launches)" % (name, 1e3 * ms / max(cnt, 1), T, cnt))
            tot += 1e3 * ms / max(cnt, 1)
    print("  kernels: %.1f us per %d frames = %.1f us per frame" % (tot, T, tot / T))
    ctx.close()

    # host -> host export, canvases against JPEG frames
    Te = 32
    host = np.stack([clip.frame(t) for t in range(16)])
    fb, cb = h * w * 3, cw * chh * 3
    cpre = lvm.LvmPreprocessParams(1, 0, 0.0, 0.0, 1.0, 1.0, 0)
    from tests_helpers_shim import c_params  # noqa
    cp = c_params(lvm, pk)
    pin, pout, pj = C.c_void_p(), C.c_void_p(), C.c_void_p()
    jcap = int(np.max(sizes) * 1.5) * Te
    assert lib.lvm_host_alloc(fb * Te, C.byref(pin)) == 0 and lib.lvm_host_alloc(cb * Te, C.byref(pout)) == 0 and lib.lvm_host_alloc(jcap, C.byref(pj)) == 0
    src = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(Te, h, w, 3))
    for i in range(Te):
        src[i] = host[i % 16]
    vp = C.c_void_p
    pi = (vp * Te)(*[pin.value + i * fb for i in range(Te)])
    pc = (vp * Te)(*[pout.value + i * cb for i in range(Te)])
    prod = (C.c_int * Te)()
    joffs = (C.c_size_t * (Te + 1))()
    for name in ("canvases", "mjpeg chunk 2", "mjpeg chunk 4", "mjpeg chunk 8", "mjpeg chunk 16"):
        if name.startswith("mjpeg"):
            os.environ["LVM_EXPORT_MJPEG_CHUNK"] = name.split()[-1]
        ex = lvm.Context(0, 1)
        ex.set_max_frames(Te)

        def call():
            if name == "canvases":
                ex._check(lib.lvm_export_frames(ex.h, C.byref(cpre), C.byref(cp), 1, Te, pi, w, h, 3, w * 3, pc, cw * 3, prod))
            else:
                ex._check(lib.lvm_export_frames_mjpeg(ex.h, C.byref(cpre), C.byref(cp), 1, Te, pi, w, h, 3, w * 3, q, pj, jcap, joffs, prod))
        for _ in range(2):
            call()
        Kx = 6
        t0 = time.perf_counter()
        for _ in range(Kx):
            call()
        dx = time.perf_counter() - t0
        down = cb if name == "canvases" else joffs[Te] / Te
        print("export host -> host, %-14s: %7.0f frames/s (%.1f us per frame; %.2f MB up + %.2f MB down per frame)" % (
            name, Kx * Te / dx, 1e6 * dx / (Kx * Te), fb / 1e6, down / 1e6))
        ex.close()
    # the decode half: 1080p frames as this repository's encoder writes them (68 restart intervals each) and as libjpeg writes them (none)
    dctx = lvm.Context(0, 1)
    dsrc = torch.from_numpy(frames).cuda()
    own8 = dctx.mjpeg_encode_device(C.c_void_p(dsrc.data_ptr()), w, h, 8, quality=90)              # the default: restart intervals of 8 MCUs
    dctx.mjpeg_set_restart_interval((w + 15) // 16)
    own = dctx.mjpeg_encode_device(C.c_void_p(dsrc.data_ptr()), w, h, 8, quality=90)               # one interval per MCU row
    dctx.mjpeg_set_restart_interval(0)
    print("restart intervals of 8 MCUs (default) against one per MCU row: %.0f KB per frame against %.0f (+%.2f %%)" % (
        np.mean([len(x) for x in own8]) / 1e3, np.mean([len(x) for x in own]) / 1e3, 100.0 * (sum(len(x) for x in own8) / sum(len(x) for x in own) - 1)))
    kinds = [("this encoder's frames, restart interval = 8 MCUs, the default (1020 lanes per frame)", own8), ("this encoder's frames, restart interval = MCU row (68 lanes per frame)", own)]
    try:
        from PIL import Image
        lj = []
        for k in range(8):
            b = io.BytesIO()
            Image.fromarray(frames[k][..., ::-1]).save(b, "JPEG", quality=90, subsampling=2)
            lj.append(b.getvalue())
        kinds.append(("libjpeg's frames (no restart markers: self-synchronising lanes of 1024 bits)", lj))
    except Exception:
        pass
    # (Huffman decoding is serial inside a restart interval: a call's time is the latency of ONE interval, whatever the number of frames)
    for label, js8 in kinds:
        for nd in (8, 64):
            js = (js8 * 8)[:nd]
            dout = torch.zeros((nd, h, w, 3), dtype=torch.uint8, device="cuda")
            blob = np.frombuffer(b"".join(js), np.uint8)
            offsn = (C.c_size_t * (nd + 1))(*np.concatenate([[0], np.cumsum([len(j) for j in js])]).tolist())
            dctx._check(lib.lvm_mjpeg_decode_device(dctx.h, blob.ctypes.data, offsn, nd, w, h, dout.data_ptr(), w * 3, w * 3 * h))
            Kd = 3
            t0 = time.perf_counter()
            for _ in range(Kd):
                dctx._check(lib.lvm_mjpeg_decode_device(dctx.h, blob.ctypes.data, offsn, nd, w, h, dout.data_ptr(), w * 3, w * 3 * h))
            dt = time.perf_counter() - t0
            print("lvm_mjpeg_decode_device, %2d x 1080p, %s: %.0f frames/s (%.1f ms per call, %.0f KB per frame)" % (
                nd, label, Kd * nd / dt, 1e3 * dt / Kd, len(blob) / nd / 1e3))
            del dout
    dctx.close()
    # file -> file: JPEG frames in, JPEG frames of the composed canvases out
    for label, ownk in (("inputs with restart intervals of 8 MCUs (this encoder's default)", own8), ("inputs with one restart interval per MCU row", own)):
        jin = (ownk * 4)[:Te]
        blob = np.frombuffer(b"".join(jin), np.uint8)
        pjin = C.c_void_p()
        assert lib.lvm_host_alloc(len(blob), C.byref(pjin)) == 0
        C.memmove(pjin, blob.ctypes.data, len(blob))
        ioffs = (C.c_size_t * (Te + 1))(*np.concatenate([[0], np.cumsum([len(j) for j in jin])]).tolist())
        os.environ["LVM_EXPORT_MJPEG_CHUNK"] = "4"
        ex = lvm.Context(0, 1)
        ex.set_max_frames(Te)
        for _ in range(2):
            ex._check(lib.lvm_export_mjpeg_frames(ex.h, C.byref(cpre), C.byref(cp), 1, Te, pjin, ioffs, w, h, q, pj, jcap, joffs, prod))
        t0 = time.perf_counter()
        for _ in range(6):
            ex._check(lib.lvm_export_mjpeg_frames(ex.h, C.byref(cpre), C.byref(cp), 1, Te, pjin, ioffs, w, h, q, pj, jcap, joffs, prod))
        dx = time.perf_counter() - t0
        print("export JPEG -> JPEG (lvm_export_mjpeg_frames), %s: %7.0f frames/s (%.1f us per frame; %.2f MB up + %.2f MB down per frame)" % (
            label, 6 * Te / dx, 1e6 * dx / (6 * Te), len(blob) / Te / 1e6, joffs[Te] / Te / 1e6))
        ex.close()
        lib.lvm_host_free(pjin)
    try:
        from PIL import Image
        im = Image.fromarray(canvas[0][..., ::-1])
        t0 = time.perf_counter()
        for _ in range(5):
            b = io.BytesIO()
            im.save(b, "JPEG", quality=q, subsampling=2)
        dt = (time.perf_counter() - t0) / 5
        print("libjpeg-turbo (Pillow %s) on one host core, the same canvas: %.1f frames/s (%.1f ms per frame, %d KB)" % (
            importlib.import_module("PIL").__version__, 1 / dt, 1e3 * dt, len(b.getvalue()) // 1000))
    except Exception as e:
        print("Pillow not available:", e)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as tests_helpers_shim
    sys.modules["tests_helpers_shim"] = tests_helpers_shim
    main()
