import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
lvm = importlib.import_module("live-video-magnification_amd")
sys.path.insert(0, "tests")
from helpers import c_params
hip = lvm.load()
T, ring, ncalls = 32, 64, 3
ck, pk = lvm.synth.config(1)
w, h = ck["w"], ck["h"]
clip = lvm.synth.Clip(**ck)
d_in = torch.stack([clip.frame_torch(t, "cuda") for t in range(ring)])
fb = w * h * 3
cp = c_params(lvm, pk)
st = torch.cuda.current_stream().cuda_stream
outs = {}
for name, env in (("c1", {"LVM_LAP_CHUNKS": "1"}), ("c2", {"LVM_LAP_CHUNKS": "2"}), ("c2u", {"LVM_LAP_CHUNKS": "2", "LVM_D0_FUSED": "0"}), ("c4", {"LVM_LAP_CHUNKS": "4"})):
    for k in ("LVM_LAP_CHUNKS", "LVM_D0_FUSED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = lvm.Context(0, 1, hip)
    ctx.set_max_frames(T)
    d_out = torch.zeros((T * ncalls, h, w, 3), dtype=torch.uint8, device="cuda")
    for call in range(ncalls):
        t = (call * T) % ring
        ctx.process_device_frames(cp, T, d_in[t].data_ptr(), w, h, 3, w * 3, fb, fb, d_out[call * T].data_ptr(), w * 3, fb, fb, st)
    torch.cuda.synchronize()
    outs[name] = d_out.cpu().numpy()
    ctx.close()
ref = outs["c1"]
for name in ("c2", "c2u", "c4"):
    d = np.abs(outs[name].astype(int) - ref.astype(int))
    bad = [int(i) for i in range(d.shape[0]) if d[i].max() > 0]
    print(name, "max diff", d.max(), "frames differing", bad[:40], len(bad))
    if bad:
        i = bad[0]; ys, xs, cs = np.nonzero(d[i]); print("  first bad frame", i, "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "count", len(ys))
