mkdir -p gpurun_out/r10
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r10/gpu_tests.txt 2>&1
tail -5 gpurun_out/r10/gpu_tests.txt
python - <<'PY'
# preprocess kernel bandwidth at 1080p (device-resident), 8 streams
import importlib, ctypes, time, numpy as np, torch
lvm = importlib.import_module("live-video-magnification_amd")
w, h, S = 1920, 1080, 8
ctx = lvm.Context(0, S)
d_in = torch.randint(0, 256, (S, h, w, 3), dtype=torch.uint8, device="cuda")
for ds, roi, gray in [(1, False, True), (2, False, False), (4, False, True), (4, True, False), (8, False, False)]:
    pre = lvm.PreprocessParams(downscale=ds, roiEnabled=roi, roiX=0.1, roiY=0.1, roiW=0.77, roiH=0.77)
    cp = lvm.to_c_preprocess(pre, gray)
    rx, ry, rw, rh, ow, oh, och = ctx.preprocess_geometry(cp, w, h, 3)
    d_out = torch.zeros((S, oh, ow, och), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: ctx.preprocess_device(cp, ctypes.c_void_p(d_in.data_ptr()), w, h, 3, w * 3, w * h * 3, ctypes.c_void_p(d_out.data_ptr()), ow * och, ow * oh * och, ctypes.c_void_p(st))
    for _ in range(5): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 200
    for _ in range(N): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
    byt = S * (rw * rh * 3 + ow * oh * och)
    print("preprocess ds=%d roi=%d gray=%d: %dx%d -> %dx%dx%d  %.1f us/launch  %.0f GB/s" % (ds, roi, gray, rw, rh, ow, oh, och, dt * 1e6, byt / dt / 1e9))
PY
