#!/usr/bin/env python3
"""bench.py -- magnified frames/s of the HIP magnification core on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode laplace|riesz|color] [--streams B]
                    [--frames-per-call T]

A "step" is one frame of the hot path for each of the B streams held by the context, over
synthetic frames that are already resident in HBM; outputs stay in HBM.  The K steps are issued
through lvm_process_device_frames in calls of T consecutive frames of the same stream(s) -- the
reference's export loop (export/Exporter.cpp:216-259) sees its frames in exactly this order; the
result of every frame is what K per-frame calls give (T = 1 selects those: the live, one-frame
latency schedule).  Default workload = BASELINE.json configs[1]: Laplace motion, 1920x1080,
6 levels, IIR 0.4-3 Hz, alpha 20, single stream, one MI355X.  With N > 1 (one rank per GPU, launched by
torch.distributed.run) every rank runs its own independent stream(s): the path shards by
stream with no data-path collective (weak scaling); RCCL is only used for the timing barrier
and the max-over-ranks reduction.

Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel, HIP-event timed on the
launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores, rank 0, N=1).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MODES = {"laplace": 1, "riesz": 2, "color": 3}  # -> BASELINE.json configs index


def level_sizes(w, h, levels):
    out = [(w, h)]
    for _ in range(levels):
        w, h = (w + 1) // 2, (h + 1) // 2
        out.append((w, h))
    return out


def kernel_alg_bytes(mode, name, w, h, ch, levels, streams, T=0):
    """Compulsory bytes one launch of kernel `name` moves (every input read once + every output
    written once) for the kernel decomposition of DESIGN.md; kernels launched once per level are
    averaged over their launches (rocprofv3 reports them under one symbol).  None = not tabulated."""
    sizes = level_sizes(w, h, levels)
    n = [a * b for a, b in sizes]
    S = streams
    P = ch * streams
    avg = lambda xs: (sum(xs) / len(xs)) if xs else None  # noqa: E731
    if mode == "laplace":
        return {
            "lap_down0": S * ch * n[0] + 4 * P * n[1],
            "lap_final": 2 * S * ch * n[0] + 4 * P * n[1],
            "pyr_down": avg([4 * P * (n[l] + n[l + 1]) for l in range(1, levels)]),
            # the wave-strip pyrDown serves the levels with >= 2^20 plane-pixels per launch (laplace.hip)
            "pyr_down_rows": avg([4 * P * (n[l] + n[l + 1]) for l in range(1, levels)
                                  if sizes[l][0] % 4 == 0 and n[l] * P >= (1 << 20)]),
            # G_l + G_{l+1} + cur_{l+1} read, hi/lo read+written, cur_l written
            "lap_up": avg([4 * P * (n[l] * 6 + 2 * n[l + 1]) for l in range(1, levels)]),
            "lap_seed": avg([4 * P * (n[l] * 3 + n[l + 1]) for l in range(1, levels)]),
        }.get(name)
    if mode == "riesz":
        nb = levels - 1
        return {
            "rz_lab": S * (3 * n[0] + 4 * n[0]),
            "rz_split": avg([4 * S * (2 * n[l] + n[l + 1]) for l in range(nb)]),
            # band 4 + prior(3) R/W 24 + phase(2) R/W 16 + registers(8) R/W 64 + amp,tc,ts 12
            "rz_phase": sum(120 * S * n[l] for l in range(nb)),          # one launch covers every band level
            "rz_seed": sum((4 + 13 * 4) * S * n[l] for l in range(nb)),
            # the register-blocked kernel serves the levels with w % 4 == 0, w >= 128, h >= 64 (riesz.hip)
            "rz_blur_amp": sum(28 * S * n[l] for l in range(nb) if sizes[l][0] % 4 == 0 and sizes[l][0] >= 128 and sizes[l][1] >= 64),
            "rz_blur_amp_small": sum(28 * S * n[l] for l in range(nb)
                                     if not (sizes[l][0] % 4 == 0 and sizes[l][0] >= 128 and sizes[l][1] >= 64)),
            "rz_collapse": avg([4 * S * (2 * n[l] + n[l + 1]) for l in range(1, nb)]),
            "rz_final": S * (6 * n[0] + 4 * n[0] + 4 * n[1]),
        }.get(name)
    if mode == "color":
        nL = n[levels]
        nV = nL * 4 ** (levels - 1)
        return {
            "col_down0": S * ch * n[0] + 4 * P * n[1],
            "pyr_down": avg([4 * P * (n[l] + n[l + 1]) for l in range(1, levels)]),
            "pyr_down_rows": avg([4 * P * (n[l] + n[l + 1]) for l in range(1, levels)
                                  if sizes[l][0] % 4 == 0 and n[l] * P >= (1 << 20)]),
            "col_append": 8 * ch * nL,
            "col_dft": 4 * P * nL * (T + 1),
            "col_norm": 8 * ch * nL,
            "pyr_up": avg([4 * P * nL * (4 ** k + 4 ** (k + 1)) for k in range(levels - 1)]),
            "col_minmax": S * ch * n[0] + 4 * P * nV,
            "col_out": 2 * S * ch * n[0] + 4 * P * nV,
        }.get(name)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--mode", default="laplace", choices=list(MODES))
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU (one launch covers all)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--levels", type=int, default=0)
    ap.add_argument("--ring", type=int, default=64, help="distinct input frames kept in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay steady-state frames from a captured hipGraph")
    ap.add_argument("--frames-per-call", type=int, default=32, help="consecutive frames of the stream(s) handed to one lvm_process_device_frames call (1 = per-frame calls)")
    ap.add_argument("--pipeline", type=int, default=0, help="cross-frame pipeline depth of lvm_process_device (0 or 1)")
    ap.add_argument("--profile-steps", type=int, default=-1, help="frames of the per-kernel timing pass (default: two calls of T frames; 0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (smoke-testing the N > 1 path)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    lvm = importlib.import_module("live-video-magnification_amd")
    cfg_idx = MODES[args.mode]
    small = None
    ck0, pk0 = lvm.synth.config(cfg_idx)
    if args.width or args.height or args.levels:
        small = (args.width or ck0["w"], args.height or ck0["h"], args.levels or pk0["levels"])
    ck, pk = lvm.synth.config(cfg_idx, small)
    w, h, levels, ch = ck["w"], ck["h"], pk["levels"], 3
    B = args.streams

    # ---- synthetic clip, staged in HBM: ring x B x h x w x 3 (stream s uses seed 1234 + s) ----
    ring = args.ring
    ids = lvm.sharding.stream_ids(rank, world, B)
    clips = [lvm.synth.Clip(seed=lvm.sharding.stream_seed(i), **{k: v for k, v in ck.items() if k != "seed"}) for i in ids]
    host = np.empty((ring, B, h, w, ch), np.uint8)
    for t in range(ring):
        for s in range(B):
            host[t, s] = clips[s].frame(t)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.empty_like(d_in)
    frame_bytes = h * w * ch

    ctx = lvm.Context(local_rank, B)
    ctx.set_graph(bool(args.graph))
    ctx.set_pipeline(args.pipeline)
    cp = lvm.LvmParams(pk["mode"], pk["levels"], pk["amplification"], pk["coWavelength"], pk["coLow"], pk["coHigh"],
                       pk["chromAttenuation"], pk["framerate"], 0)
    stream = torch.cuda.current_stream().cuda_stream

    in_ptrs = [d_in[t].data_ptr() for t in range(ring)]
    out_ptrs = [d_out[t].data_ptr() for t in range(ring)]
    fast = ctx.make_stepper(cp, w, h, ch, w * ch, frame_bytes, w * ch, frame_bytes, stream)

    def step(i):
        t = i % ring
        rc = fast(in_ptrs[t], out_ptrs[t])
        if rc != 0:
            ctx._check(rc)

    # temporal batches: one call = T consecutive ring frames (frame stride = B * frame_bytes)
    T = max(1, min(args.frames_per_call, ring))
    fn_frames = ctx.lib.lvm_process_device_frames
    import ctypes as C
    prod_arr = (C.c_int * T)()
    p_ref = C.byref(cp)

    def run_frames(first, count):
        """process `count` frames starting at global frame index `first`, in calls of at most T frames
        that never wrap around the ring"""
        i = first
        end = first + count
        while i < end:
            t = i % ring
            nf = min(T, end - i, ring - t)
            if nf == 1:
                step(i)
            else:
                rc = fn_frames(ctx.h, p_ref, nf, in_ptrs[t], w, h, ch, w * ch, frame_bytes, frame_bytes * B, out_ptrs[t], w * ch,
                               frame_bytes, frame_bytes * B, prod_arr, stream)
                if rc != 0:
                    ctx._check(rc)
            i += nf

    red_dev = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else torch.device("cpu")
    if args.mode == "color":   # the rolling window must be full before the steady state starts
        args.warmup = max(args.warmup, lvm.load().lvm_optimal_buffer_size(int(pk["framerate"])) + 32)
    n = 0
    run_frames(0, args.warmup); n += args.warmup
    base = n
    if T > 1:
        # K steps (frames) issued as ceil(K / T) batched calls; timed_steps sees it as one "step" of K frames
        dt = lvm.sharding.timed_steps(lambda i: run_frames(base, args.steps), 1, dist, torch.cuda.synchronize,
                                      red_dev, finish=lambda: ctx.flush(stream))
    else:
        dt = lvm.sharding.timed_steps(lambda i: step(base + i), args.steps, dist, torch.cuda.synchronize, torch.device("cuda", local_rank),
                                      finish=lambda: ctx.flush(stream))
    n += args.steps
    fps = lvm.sharding.aggregate_fps(world, B, args.steps, dt)

    # ---- per-kernel timing pass (HIP events on the launch stream) -> roofline ----
    roofline = None
    kernels = {}
    if args.profile_steps < 0:
        args.profile_steps = 2 * T if T > 1 else 60        # whole calls only: every launch then covers exactly T frames
    if rank == 0 and args.profile_steps > 0:
        ctx.flush(stream)
        ctx.set_pipeline(0)           # per-kernel event timing uses the plain schedule
        pad = (-n) % ring             # start on a ring boundary so that every call of the pass covers exactly T frames
        run_frames(n, pad); n += pad
        ctx.profile(True)
        run_frames(n, args.profile_steps); n += args.profile_steps
        torch.cuda.synchronize()
        prof = ctx.profile_collect()
        ctx.profile(False)
        tot = sum(v[0] for v in prof.values()) or 1.0
        Twin = lvm.load().lvm_optimal_buffer_size(int(pk["framerate"]))
        for name, (ms, cnt) in prof.items():
            avg_us = 1e3 * ms / max(cnt, 1)
            # a launch of the temporally batched schedule covers T_frames frames of every stream
            # frames one launch covers: T, except that the colour mode cuts a call into chunks of <= 32 frames (window ring)
            T_launch = min(T, 32) if args.mode == "color" else T
            ab = kernel_alg_bytes(args.mode, name, w, h, ch, levels, B * T_launch, Twin)
            kernels[name] = {"avg_us": round(avg_us, 3), "launches": cnt, "share": round(ms / tot, 4),
                             "alg_bytes": ab, "gbs": (round(ab / (avg_us * 1e-6) / 1e9, 1) if ab else None)}
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0]
        k = kernels[dom]
        # HBM bytes per launch from the PMC counters (separate rocprofv3 passes, committed under profiles/)
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_%s.json" % args.mode)))
            if pm.get("key") == "%s|%dx%d|L%d|B%d|T%d" % (args.mode, w, h, levels, B, T):
                traffic = pm["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        if k["gbs"]:
            roofline = {"bound": "hbm", "kernel": dom, "achieved": k["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(k["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "avg_us": k["avg_us"], "alg_bytes_per_launch": k["alg_bytes"]}
            # what actually limits that kernel, from the committed SQ-counter passes (profiles/*_sq_counters.txt)
            limiter = {"lap_final": "VALU-bound (Lab arithmetic): 77 % VALU issue utilisation, 31 % LDS",
                       "lap_down0": "VALU-bound (Lab arithmetic): 68 % VALU issue utilisation",
                       "rz_blur_amp": "39 % VALU, 51 % LDS busy; 3.2 TB/s of compulsory traffic",
                       "rz_final": "49 % VALU, 57 % LDS busy (9x9 taps + table lookups)",
                       "rz_phase": "VALU-bound: 79 % VALU issue utilisation (acosf, sqrt/div, float64 filter products)"}.get(dom)
            if limiter:
                roofline["limiter"] = limiter
            # the committed rocprofv3 --stats summary of this command (kernels back to back, no event gaps): its average for
            # the same kernel.  The VALU-bound kernels run a few % slower there (sustained clocks); see profiles/README.md.
            try:
                sym = {"lap_final": "ILb1ELb0EEEvPKhllPhlliiPKfiiNS_7LabCoefEfiiiiPf", "lap_down0": "down0_rowsILb1ELb0", "lap_up": "k_lap_upILb0ELi1"}.get(dom)
                if sym and args.mode == "laplace" and traffic is not None:
                    for line in open(os.path.join(ROOT, "profiles", "r01_rocprof_laplace_kernel_stats.txt")):
                        if sym in line:
                            roofline["rocprof_avg_us"] = float(line.split()[2])
                            break
            except Exception:
                pass
    b_alg = lvm.load().lvm_algorithmic_bytes(pk["mode"], w, h, ch, levels, pk["framerate"])
    frame_frac = b_alg * (fps / world) / (HBM_PEAK_GBS * 1e9)

    # ---- CPU baseline: the oracle (CPU restatement of the reference) on this box's cores ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        o = po.Oracle()
        P = po.make_params(**pk)
        nthreads = max(1, min(16, os.cpu_count() or 1))   # beyond ~16 threads fork/join over small levels dominates
        po.lib().lvmo_set_threads(nthreads)
        frames = [host[t % ring, 0] for t in range(4)]
        for f in frames[:2]:
            o.process(f, P)                       # first frame seeds, second warms caches
        tc0 = time.perf_counter(); cnt = 0
        while time.perf_counter() - tc0 < 12.0 and cnt < 400:
            o.process(host[cnt % ring, 0], P); cnt += 1
        cdt = time.perf_counter() - tc0
        cpu = {"value": round(cnt / cdt, 3), "unit": "frames/s", "cores": nthreads, "kind": "port",
               "sample": "%d frames of the same %dx%d L%d %s clip, CPU oracle (restatement of the reference; "
                         "OpenCV unavailable), OpenMP over rows" % (cnt, w, h, levels, args.mode)}

    if rank == 0:
        out = {
            "metric": "magnified frames/sec at 1080p, Laplace-motion 6 levels; % HBM roofline" if args.mode == "laplace"
                      else "magnified frames/sec (%s)" % args.mode,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 5),
            "host_enqueue_ms_per_step": round(1e3 * getattr(lvm.sharding.timed_steps, "host_seconds", 0.0) / args.steps, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d levels, %d stream(s)/GPU, device-resident u8 BGR in/out" %
                                   (args.mode, w, h, levels, B),
                       "baseline_config": cfg_idx, "streams_per_gpu": B, "ring_frames": ring,
                       "frames_per_call": T, "pipeline_depth": args.pipeline, "hip_graph": bool(args.graph)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "frame_alg_bytes": b_alg, "frame_roofline_frac": round(frame_frac, 5),
            "kernels": kernels,
        }
        print(json.dumps(out))
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
