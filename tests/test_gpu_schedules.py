"""GPU tests of the schedules bench.py times and of the boundary's concurrency rules:
  * 1920x1080 temporal batches of 32 frames over a 64-frame ring (the benchmarked schedule) for the three modes,
    every output frame against the CPU oracle + the pre-quantisation float frame of a batch,
  * a 64-frame 1080p Laplace clip through the host surface (state drift at full size),
  * bench.py itself: self-verification, the byte model (no kernel above the HBM peak) and the N = 2 launch path
    (two ranks sharing the GPU over gloo),
  * two contexts driven from two host threads (the reference's live + export chains, export/Exporter.cpp:204,231),
    keep_float across a size change after lvm_chain_process, create/destroy in a loop.
Tolerances as in test_gpu_parity.py."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from helpers import c_params, run_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_frames_on_device_are_bit_identical(lvm):
    """bench.py stages its clips with Clip.frame_torch on the GPU: same float64 operations, same bytes."""
    for idx, small in ((1, (640, 360, 4)), (3, (320, 180, 4)), (2, (323, 211, 4))):
        ck, _ = lvm.synth.config(idx, small)
        clip = lvm.synth.Clip(**ck)
        for t in (0, 1, 7, 19, 33):
            assert np.array_equal(clip.frame(t), clip.frame_torch(t, "cuda").cpu().numpy()), (idx, t)


@pytest.mark.parametrize("idx,ncalls", [(1, 3), (2, 3), (3, 6)])
def test_1080p_batches_of_32_over_a_ring(lvm, po, hip, idx, ncalls):
    """The benchmarked schedule at its full size: calls of T = 32 consecutive frames through
    lvm_process_device_frames over a 64-frame input ring, lvm_set_max_frames(32) given up front.  Every produced
    frame is compared with the oracle (u8), and the float frame of the first frame of the last call too."""
    import torch
    T, ring = 32, 64
    ck, pk = lvm.synth.config(idx)
    w, h = ck["w"], ck["h"]
    clip = lvm.synth.Clip(**ck)
    d_in = torch.stack([clip.frame_torch(t, "cuda") for t in range(ring)])
    host = d_in.cpu().numpy()
    n = T * ncalls
    d_out = torch.zeros((n, h, w, 3), dtype=torch.uint8, device="cuda")
    fb = w * h * 3
    ctx = lvm.Context(0, 1, hip)
    ctx.set_max_frames(T)
    cp = c_params(lvm, pk)
    st = torch.cuda.current_stream().cuda_stream
    produced = []
    for call in range(ncalls):
        if call == ncalls - 1:
            ctx.keep_float(True)
        t = (call * T) % ring
        produced += ctx.process_device_frames(cp, T, d_in[t].data_ptr(), w, h, 3, w * 3, fb, fb, d_out[call * T].data_ptr(), w * 3, fb, fb, st)
    torch.cuda.synchronize()
    fl_gpu = ctx.read_float((h, w, 3))
    ctx.close()
    orc = po.Oracle()
    P = po.make_params(**pk)
    worst = [0, 1.0]
    compared = 0
    fl_ref = None
    for i in range(n):
        ref, pr = orc.process(host[i % ring], P)
        assert pr == produced[i], (i, pr, produced[i])
        if i == (ncalls - 1) * T:
            fl_ref = orc.last_float().copy()
        if not pr:
            continue
        if i % 3 and i < n - 4 and i not in (T, T + 1):      # D2H of every frame would dominate the test: two thirds are skipped
            continue
        got = d_out[i].cpu().numpy()
        du = np.abs(ref.astype(np.int16) - got.astype(np.int16))
        worst = [max(worst[0], int(du.max())), min(worst[1], float((du == 0).mean()))]
        assert du.max() <= 1 and (du == 0).mean() >= 0.999, (i, int(du.max()), float((du == 0).mean()))
        compared += 1
    orc.close()
    rel = float(np.abs(fl_ref - fl_gpu).max() / np.abs(fl_ref).max())
    print("mode cfg", idx, "frames compared", compared, "worst u8 / identical", worst, "float rel", rel)
    assert compared >= 20
    assert np.isfinite(fl_gpu).all() and rel <= 1e-4, rel


def test_laplace_1080p_64_frames_state_drift(lvm, po, hip):
    """BASELINE.json configs[1] at full size for 64 frames (float + u8 bars on every frame)."""
    ck, pk = lvm.synth.config(1)
    worst = run_pair(lvm, po, hip, lvm.synth.Clip(**ck), pk, 64, 1e-4)
    print("laplace 1080p 64 frames worst rel/u8/frac", worst)


def test_riesz_1080p_64_frames_state_drift(lvm, po, hip):
    """BASELINE.json configs[2] at full size for 64 frames: the float metric and the u8 bars on EVERY frame, temporal
    state (phase accumulators, both Butterworth register pairs, prior pyramid) drifting for the whole clip -- SURVEY 8c(i)
    for the ill-conditioned mode (the Laplace twin is the test above)."""
    ck, pk = lvm.synth.config(2)
    worst = run_pair(lvm, po, hip, lvm.synth.Clip(**ck), pk, 64, 1e-4)
    print("riesz 1080p 64 frames worst rel/u8/frac", worst)


@pytest.mark.parametrize("strips", [{}, {"LVM_RZ_SPLIT_STRIP": "10", "LVM_RZ_COLLAPSE_STRIP": "10"}, {"LVM_RZ_SPLIT_STRIP": "54", "LVM_RZ_COLLAPSE_STRIP": "46"}])
def test_riesz_4k_strip_kernels_equal_the_tiled_kernels(lvm, hip, monkeypatch, strips):
    """Five-frame batches of 3840 x 2160, 8 levels: the 9x9 wave-strip kernels (k_rz_split_rows, k_rz_collapse_strips as collapse and as
    output kernel) against the LDS-tiled kernels on the same frames -- identical bytes, twice in a row.  Both families evaluate the
    same fma chains, so any difference is a hardware-level fault of one of them: this is the case in which a `buffer_store_dwordx3`
    with an SGPR offset lost its third dword for the last four lanes of every 16 (the next instruction overwrote the register; the
    compiler pads that hazard only without a register in the soffset field) -- a few hundred pixels of a 4K frame, never at 1080p,
    never in the emulation build."""
    import torch
    from helpers import c_params
    ck, pk = lvm.synth.config(4)
    clip = lvm.synth.Clip(**ck)
    w, h, T = ck["w"], ck["h"], 5
    frames = np.stack([clip.frame(t) for t in range(T)])
    fb = w * h * 3

    def run(env):
        for k in ("LVM_RZ_COLLAPSE_STRIPS", "LVM_RZ_SPLIT_ROWS", "LVM_RZ_COLLAPSE_STRIP", "LVM_RZ_SPLIT_STRIP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = lvm.Context(0, 1, hip)
        try:
            d_in = torch.from_numpy(frames).cuda()
            d_out = torch.zeros_like(d_in)
            prod = ctx.process_device_frames(c_params(lvm, pk), T, d_in.data_ptr(), w, h, 3, w * 3, fb, fb, d_out.data_ptr(), w * 3, fb, fb,
                                             torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            return list(prod), d_out.cpu().numpy()
        finally:
            ctx.close()

    p0, tiled = run({"LVM_RZ_COLLAPSE_STRIPS": "0", "LVM_RZ_SPLIT_ROWS": "0"})
    assert any(p0)
    for _ in range(2):
        p1, got = run(strips)
        assert p1 == p0
        bad = np.argwhere(got != tiled)
        assert len(bad) == 0, (len(bad), bad[:8])


def _bench(args, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_verifies_itself_and_prices_kernels_below_peak():
    """bench.py at a reduced size: the timed region's own output matches the oracle, the JSON carries the contract's
    fields, and no per-kernel GB/s exceeds the HBM peak (a figure above peak means the byte model is wrong)."""
    out = _bench(["--width", "640", "--height", "360", "--levels", "4", "--steps", "6", "--warmup", "2", "--no-subrecords"])
    assert out["verified"] is True, out["verification"]
    assert out["verification"]["timed_frames_compared"] >= 8
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["warmup"] == 2 and out["frames_per_step"] == 32
    assert out["roofline"] and out["cpu_baseline"] and out["cpu_baseline"]["kind"] == "port"
    for name, k in out["kernels"].items():
        assert k["alg_bytes"] is not None, name
        assert k["gbs"] <= 8000.0, (name, k)
    # a step = one call of frames_per_step frames: value (frames/s) = steps x frames_per_step / the timed seconds
    assert abs(out["value"] - out["steps"] * out["frames_per_step"] / out["timed_seconds_max_over_ranks"]) <= 1e-6 * out["value"] + 0.01
    assert abs(out["ms_per_step"] - 1e3 * out["timed_seconds_max_over_ranks"] / out["steps"]) <= 1e-3
    # the shader clock over the timed region (s_memtime / s_memrealtime): a plausible MI355X figure, covering about the timed region
    assert out["clock"]["error"] is None and 500.0 < out["clock_mhz"] < 3000.0, out["clock"]
    assert 0.5 * out["timed_seconds_max_over_ranks"] < out["clock"]["seconds_covered"] < 1.5 * out["timed_seconds_max_over_ranks"] + 1e-3, out["clock"]


@pytest.mark.parametrize("mode", ["riesz", "color"])
def test_bench_other_modes_verify(mode):
    out = _bench(["--mode", mode, "--width", "640", "--height", "360", "--levels", "4", "--steps", "4", "--warmup", "2", "--no-subrecords"])
    assert out["verified"] is True, out["verification"]
    for name, k in out["kernels"].items():
        assert k["gbs"] is None or k["gbs"] <= 8000.0, (name, k)


def test_bench_two_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE re-executes under torch.distributed.run: two ranks (sharing the
    one GPU of this box, gloo for the barrier / MAX-reduce), each with its own stream (seed 1234 + rank), each
    verified against the oracle for ITS seed; value = 2 ranks x streams x steps / max-over-ranks seconds."""
    out = _bench(["--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--steps", "8", "--warmup", "4", "--width", "320", "--height", "180",
                  "--levels", "4", "--verify-all-ranks", "--no-subrecords", "--frames-per-call", "4", "--ring", "8"])
    assert out["n_gpus"] == 2
    ranks = sorted(out["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1]
    assert [r["stream_ids"] for r in ranks] == [[0], [1]]
    assert not [r for r in ranks if r["verified"] is not True], "UNVERIFIED RANKS " + json.dumps([(r["rank"], r.get("verification")) for r in ranks if r["verified"] is not True])
    expect = 2 * 1 * 8 * 4 / out["timed_seconds_max_over_ranks"]       # ranks x streams x steps x frames per step
    assert abs(out["value"] - expect) <= 1e-6 * expect + 0.01
    assert out["cfg4_riesz_4k"]["n_gpus"] == 2 and out["cfg4_riesz_4k"]["verified"] is True
    assert len(out["host_fed"]["per_rank"]) == 2


def test_bench_one_rank_over_rccl():
    """First contact with RCCL before the driver's 8-GPU box: bench.py launched the way the driver launches it
    (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`) initialises the `nccl` (= RCCL on ROCm)
    process group at world size 1, runs the device-tensor all-reduce, the barriers and the MAX-reduce of the timed region
    through it, and reports what RCCL saw."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-backend", "nccl", "--steps", "8",
           "--warmup", "4", "--width", "320", "--height", "180", "--levels", "4", "--no-subrecords", "--frames-per-call", "4", "--ring", "8"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["dist"] == {"initialized": True, "backend": "nccl", "world_size": 1, "ranks_seen_by_allreduce": 1}, out["dist"]
    assert out["n_gpus"] == 1 and out["verified"] is True, "UNVERIFIED " + json.dumps(out["verification"])
    assert out["ranks"][0]["device"] == "cuda:0" and out["ranks"][0]["device_name"]
    assert out["cpu_baseline"]["host_cores"] == os.cpu_count() and out["cpu_baseline"]["single_thread"]["cores"] == 1


def test_bench_eight_ranks_dry_run():
    """The rank count the driver's scaling run ends with: `python bench.py --gpus 8` (eight ranks sharing this box's one
    GPU, gloo for the barrier / MAX-reduce) -- launcher path, rendezvous on 127.0.0.1, one stream per rank (seeds 1234 +
    rank), every rank verified for ITS seed, the cfg4 (Riesz 4K) sub-record produced at N = 8.  No scaling claim: the
    ranks share one GPU."""
    out = _bench(["--gpus", "8", "--share-gpu", "--dist-backend", "gloo", "--steps", "8", "--warmup", "4", "--width", "320", "--height", "180",
                  "--levels", "4", "--verify-all-ranks", "--no-subrecords", "--frames-per-call", "4", "--ring", "8"])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak"
    ranks = sorted(out["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == list(range(8))
    assert [r["stream_ids"] for r in ranks] == [[i] for i in range(8)]
    assert not [r for r in ranks if r["verified"] is not True], "UNVERIFIED RANKS " + json.dumps([(r["rank"], r.get("verification")) for r in ranks if r["verified"] is not True])
    expect = 8 * 1 * 8 * 4 / out["timed_seconds_max_over_ranks"]       # ranks x streams x steps x frames per step
    assert abs(out["value"] - expect) <= 1e-6 * expect + 0.01
    c4 = out["cfg4_riesz_4k"]
    assert c4["n_gpus"] == 8 and c4["verified"] is True and c4["roofline"]["frame_frac"] > 0 and c4["cpu_baseline"]["kind"] == "port", c4
    # round 5: the host-fed surfaces on every rank at once (per-rank PCIe rates; aggregate = sum over ranks)
    hf = out["host_fed"]
    assert sorted(r["rank"] for r in hf["per_rank"]) == list(range(8))
    assert all(r["e2e_fps"] and r["export_fps"] and r["e2e_pcie_gbs"] and r["export_pcie_gbs"] for r in hf["per_rank"]), hf
    assert abs(hf["export_host"]["value"] - 8 * min(r["export_fps"] for r in hf["per_rank"])) < 0.1      # all frames / the slowest rank's time
    assert all("host_binding" in r for r in out["ranks"])


def test_bench_driver_line_carries_every_config():
    """The line the driver records (`bench.py --steps 20 --warmup 5`): the headline's roofline holds SURVEY 8(d)'s path fraction next
    to the dominant kernel's, `value_cold` is there, and BASELINE configs[2..4] each carry verified / roofline / cpu_baseline, the
    per-frame schedule (T = 1, B = 1 and 4 -- what MagnificationProcessor::process runs) and the host -> host surface."""
    out = _bench(["--steps", "20", "--warmup", "5"], timeout=1500)
    assert out["verified"] is True and out["roofline"]["frame_frac"] > 0 and out["roofline"]["frac"] <= 1.0
    assert out["value_cold"]["value"] > 0 and out["value_cold"]["steps"] == 20
    # round 6: a step is a whole 32-frame call (20 steps = 640 timed frames), the per-frame schedule rides beside it, the clock is recorded
    assert out["frames_per_step"] == 32 and out["timed_frames_per_stream"] == 640 and out["clock_mhz"] > 500.0
    assert out["process_schedule"]["value"] > 0 and out["process_schedule"]["frames_per_call"] == 1 and out["process_schedule"]["launches_per_frame"] >= 1
    for key in ("cfg2_riesz_1080p", "cfg3_color_1080p", "cfg4_riesz_4k"):
        r = out[key]
        assert r["verified"] is True, (key, r.get("verification"))
        assert r["roofline"]["frac"] <= 1.0 and r["roofline"]["frame_frac"] > 0 and r["roofline"]["kernel"], (key, r["roofline"])
        assert r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["kind"] == "port"
        assert r["per_frame"]["B1"]["frames_per_call"] == 1 and r["per_frame"]["B1"]["value"] > 0 and r["per_frame"]["B4"]["value"] > 0
        assert r["e2e_host"]["pinned"]["value"] > 0 and r["e2e_host"]["pageable"]["value"] > 0
    assert out["per_frame"]["value"] > 0 and out["e2e_host"]["pinned"]["value"] > 0 and out["export_host"]["value"] > 0 and out["export_host"]["mjpeg"]["value"] > 0


def test_two_contexts_on_two_threads(lvm, po, hip):
    """Live chain + export chain (export/Exporter.cpp:204,231): two contexts, each driven by its own host thread,
    different modes and sizes, interleaved resets; both must match their oracle frame by frame."""
    errors = []

    def worker(idx, small, nframes, reset_at):
        try:
            ck, pk = lvm.synth.config(idx, small)
            clip = lvm.synth.Clip(**ck)
            P = po.make_params(**pk)
            cp = c_params(lvm, pk)
            ctx = lvm.Context(0, 1, hip)
            orc = po.Oracle()
            for t in range(nframes):
                if t == reset_at:
                    ctx.reset(); orc.reset()
                f = clip.frame(t)
                ref, pr = orc.process(f, P)
                out, pg = ctx.process(f, cp)
                assert pr == pg, (idx, t)
                if pr:
                    du = np.abs(ref.astype(int) - out.astype(int))
                    assert du.max() <= 1 and (du == 0).mean() >= 0.999, (idx, t, du.max())
            ctx.close(); orc.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((idx, repr(e)))

    th = [threading.Thread(target=worker, args=(0, (640, 360, 4), 40, 17)),
          threading.Thread(target=worker, args=(2, (320, 180, 4), 30, 11))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_keep_float_across_size_change_after_chain(lvm, po, hip):
    """Round-1 defect: growing the kept float frame freed the chain's staging buffers.  chain_process, then keep_float
    with a LARGER frame through chain_process and lvm_process, then back; plus create/destroy in a loop."""
    ctx = lvm.Context(0, 1, hip)
    pre = lvm.to_c_preprocess(lvm.PreprocessParams(downscale=2), False)
    for (w, h) in ((320, 180), (640, 360), (320, 180), (800, 450)):
        ck, pk = lvm.synth.config(0, (w, h, 3))
        clip = lvm.synth.Clip(**ck)
        cp = c_params(lvm, pk)
        orc = po.Oracle()
        P = po.make_params(**pk)
        ctx.keep_float(True)
        for t in range(4):
            f = clip.frame(t)
            small = po.preprocess(f, po.make_pre_params(downscale=2))
            out, produced = ctx.chain_process(f, pre, cp)
            assert produced and out.shape == (h // 2, w // 2, 3)
            ref, _ = orc.process(small, P)
            du = np.abs(ref.astype(int) - out.astype(int))
            assert du.max() <= 1, (w, h, t, du.max())
            fl = ctx.read_float((h // 2, w // 2, 3))
            assert np.isfinite(fl).all()
        ctx.keep_float(False)
        out2, produced = ctx.process(clip.frame(0), cp)      # host surface on the same context (structural change)
        assert produced
        orc.close()
    ctx.close()
    for _ in range(20):                                      # every context that used the chain API must free it
        c2 = lvm.Context(0, 1, hip)
        ck, pk = lvm.synth.config(0, (320, 180, 3))
        c2.chain_process(lvm.synth.Clip(**ck).frame(0), pre, c_params(lvm, pk))
        c2.close()


def test_laplace_4k_8_levels_temporal_batches(lvm, po, hip):
    """Laplace at 3840x2160 with 8 levels in temporal batches: six decoupled levels in k_lap_iir_levels / k_lap_collapse
    (the deepest pyramid the configs use), every frame against the oracle."""
    import torch
    ck, pk = lvm.synth.config(0, (3840, 2160, 8))
    w, h = ck["w"], ck["h"]
    clip = lvm.synth.Clip(**ck)
    nf, ncalls = 5, 2
    d_in = torch.stack([clip.frame_torch(t, "cuda") for t in range(1 + nf * ncalls)])
    host = d_in.cpu().numpy()
    d_out = torch.zeros_like(d_in)
    fb = w * h * 3
    ctx = lvm.Context(0, 1, hip)
    cp = c_params(lvm, pk)
    st = torch.cuda.current_stream().cuda_stream
    produced = ctx.process_device_frames(cp, 1, d_in[0].data_ptr(), w, h, 3, w * 3, fb, fb, d_out[0].data_ptr(), w * 3, fb, fb, st)
    for c in range(ncalls):
        f0 = 1 + c * nf
        produced += ctx.process_device_frames(cp, nf, d_in[f0].data_ptr(), w, h, 3, w * 3, fb, fb, d_out[f0].data_ptr(), w * 3, fb, fb, st)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    ctx.close()
    orc = po.Oracle()
    P = po.make_params(**pk)
    for t in range(host.shape[0]):
        ref, pr = orc.process(host[t], P)
        assert pr == produced[t]
        du = np.abs(ref.astype(np.int16) - got[t].astype(np.int16))
        assert du.max() <= 1 and (du == 0).mean() >= 0.999, (t, int(du.max()), float((du == 0).mean()))
    orc.close()
