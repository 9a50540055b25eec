mkdir -p gpurun_out/r14
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python - <<'PY'
import importlib, time, numpy as np
lvm = importlib.import_module("live-video-magnification_amd")
ck, pk = lvm.synth.config(1)
clip = lvm.synth.Clip(**ck)
frames = [clip.frame(t) for t in range(8)]
from tests.helpers import c_params
for mode_idx, name in ((1, "laplace"), (2, "riesz"), (3, "color")):
    ck, pk = lvm.synth.config(mode_idx)
    cp = c_params(lvm, pk)
    ctx = lvm.Context(0, 1)
    for t in range(40 if mode_idx != 3 else 140): ctx.process(frames[t % 8], cp)
    N = 200
    t0 = time.perf_counter()
    for t in range(N): ctx.process(frames[t % 8], cp)
    dt = (time.perf_counter() - t0) / N
    print("lvm_process host path %s 1080p: %.1f fps (%.3f ms per frame, 12.4 MB over PCIe)" % (name, 1 / dt, dt * 1e3))
    ctx.close()
# chain with downscale 2
pre = lvm.to_c_preprocess(lvm.PreprocessParams(downscale=2), False)
ck, pk = lvm.synth.config(1); cp = c_params(lvm, pk)
ctx = lvm.Context(0, 1)
for t in range(40): ctx.chain_process(frames[t % 8], pre, cp)
t0 = time.perf_counter()
for t in range(200): ctx.chain_process(frames[t % 8], pre, cp)
dt = (time.perf_counter() - t0) / 200
print("lvm_chain_process downscale 2 laplace: %.1f fps (%.3f ms)" % (1 / dt, dt * 1e3))
PY
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 64 --warmup 32 --share-gpu --dist-backend gloo --no-cpu-baseline 2>&1 | tail -2
