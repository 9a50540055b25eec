#!/usr/bin/env python3
"""bench.py -- magnified frames/s of the HIP magnification core on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode laplace|riesz|color] [--streams B]
                    [--frames-per-call T] [--no-verify] [--no-subrecords]

A "step" is ONE PASS of the hot path over one batch: one lvm_process_device_frames call of T = --frames-per-call
consecutive frames (default 32) of each of the B streams of the context, over synthetic frames that are already
resident in HBM; outputs stay in HBM.  K steps = K * T frames per stream -- the production shape of the batched
surface: the reference's export loop (export/Exporter.cpp:216-259) sees its frames in exactly this order, and every
frame's result is what per-frame calls give.  `value` stays in frames/s, `frames_per_step` = T, `ms_per_step` = one call.
(Until round 5 a step was a frame, so that `--steps 20` timed ONE 20-frame call, fill and drain included.)  The
schedule MagnificationProcessor::process itself runs -- T = 1, one stream -- is reported beside it as
`process_schedule`.  Headline workload = BASELINE.json configs[1]: Laplace motion, 1920x1080, 6 levels, IIR 0.4-3 Hz,
alpha 20, one stream per GPU.

Sequence of one run (global frame index i reads input ring slot i % ring and writes output slot i):
    priming   untimed, the SAME call shape as the timed region (first frame seeds, then whole calls of
              T frames; colour mode: until the rolling window is full) -- every buffer the timed
              calls need exists afterwards (lvm_set_max_frames sizes the batch arenas when the state is created)
    warmup    W untimed steps (calls)
    probe     one more call with the float frame kept (parity metric (i) of SURVEY.md 8c), then the clock ramp
    timed     EXACTLY K steps, bracketed by barrier + device synchronisation, MAX over ranks
    clock     the same K steps once more with a one-lane kernel on a second stream reading the shader-clock and the
              100 MHz counters at both ends -> `clock_mhz` (behind the timed region: the probe's wave costs 0-10 %)
    verify    rank 0 (N = 1; every rank with --verify-all-ranks): the CPU oracle replays the same frames from
              frame 0 through the first <= 12 timed calls, and the bench's OWN output frames of those calls -- >= 8 --
              are compared with it (u8 <= 1 LSB and >= 99.9 % identical; float probe <= 1e-4 relative).  The replay
              doubles as the cpu_baseline sample (~20 s of CPU work whatever K is).
    profile   per-kernel HIP-event pass (two whole calls) -> roofline of the dominant kernel
    sub-records (N = 1 unless --subrecords): per-frame schedule (T = 1), host-to-host lvm_process (pageable and
              pinned frames), B = 4 / 16 streams per launch, BASELINE configs[4] (Riesz 3840x2160 L8, one stream per
              GPU; measured at every N), and configs[2] / configs[3] (Riesz / Color 1080p, verified against the oracle)

With --gpus N > 1 and no WORLD_SIZE in the environment the script re-executes itself under
torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).  Every rank runs its own independent
stream(s) (seeds 1234 + global stream id): the path shards by stream with no data-path collective (weak
scaling); RCCL only carries the timing barrier and the MAX-reduce of the elapsed time.

Prints ONE JSON line (rank 0) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import importlib
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MODES = {"laplace": 1, "riesz": 2, "color": 3}  # -> BASELINE.json configs index
PROFILE_ROUND = "r06"


def level_sizes(w, h, levels):
    out = [(w, h)]
    for _ in range(levels):
        w, h = (w + 1) // 2, (h + 1) // 2
        out.append((w, h))
    return out


def kernel_alg_bytes(mode, name, w, h, ch, levels, S, T, Twin=0):
    """Compulsory bytes ONE launch of the kernel reported as `name` moves when it covers T consecutive frames of S
    streams: every per-frame input read once and every per-frame output written once (x T), every temporal state
    word read once and written once PER LAUNCH (the batched kernels keep the state in registers across their T
    frames).  Per-level launches carry their level in the name ("lap_up_l1").  None = not tabulated."""
    sizes = level_sizes(w, h, levels)
    n = [a * b for a, b in sizes]
    P = ch * S
    lvl = None
    base = name
    if "_l" in name and name.rsplit("_l", 1)[1].isdigit():
        base, l_s = name.rsplit("_l", 1)
        lvl = int(l_s)
    # Round 3: 3-channel frames pass through OpenCV's forward Lab table ONCE (labconv.hip) and the kernels that need
    # Lab(in) read integer planes back: 2 bytes (iL) + 4 bytes (ia | ib << 16) per pixel instead of 3 bytes of BGR.
    lab_in = 6 if ch == 3 else ch
    if base == "lab_lut":
        return T * S * n[0] * (3 + (6 if mode == "laplace" else 8))              # Riesz: L as float plane
    if mode == "laplace":
        if base == "lap_down0_lut":          # table conversion fused into the first kernel: BGR in, integer planes + G_1 out
            return T * (S * (3 + 6) * n[0] + 4 * P * n[1])
        if base == "lap_down0":
            return T * (S * lab_in * n[0] + 4 * P * n[1])
        if base == "lap_final":
            return T * (S * (lab_in + ch) * n[0]) + T * 4 * P * n[1]          # planes in, BGR out, + cur_1 read
        if base in ("pyr_down", "pyr_down_rows") and lvl is not None:
            return T * 4 * P * (n[lvl] + n[lvl + 1])
        if base == "pyr_down2" and lvl is not None:
            return T * 4 * P * (n[lvl] + n[lvl + 1] + n[lvl + 2])
        if base == "pyr_down3" and lvl is not None:
            return T * 4 * P * (n[lvl] + n[lvl + 1] + n[lvl + 2] + n[lvl + 3])
        if base == "lap_up" and lvl is not None:
            per_frame = 4 * P * (n[lvl] + n[lvl + 1] + (n[lvl + 1] if lvl + 1 <= levels - 1 else 0) + n[lvl])
            return T * per_frame + 16 * P * n[lvl]                            # hi/lo read + written once per launch
        # round 6: the IIR + collapse launches start at level F = 3 when the pyramid has >= 5 levels (level 2 is a fused lap_up step then)
        F = 3 if (levels >= 5 and T >= 4 and os.environ.get("LVM_LAP_SPLIT_FROM", "3") != "2") else 2
        if base == "lap_iir":        # levels F .. L-1 in one launch: G_l, G_{l+1} read, m_l written per frame; states once
            return sum(T * 4 * P * (2 * n[l] + n[l + 1]) + 16 * P * n[l] for l in range(F, levels))
        if base == "lap_collapse":   # m_F .. m_{L-1} read, cur_F written
            return T * 4 * P * (sum(n[l] for l in range(F, levels)) + n[F])
        if base == "lap_seed" and lvl is not None:
            return 4 * P * (n[lvl] * 3 + n[lvl + 1])
        return None
    if mode == "riesz":
        nb = levels - 1
        big = lambda l: sizes[l][0] % 4 == 0 and sizes[l][0] >= 128 and sizes[l][1] >= 64  # noqa: E731
        if base == "rz_lab":
            return T * S * (3 * n[0] + 4 * n[0])
        if base == "rz_split" and lvl is not None:
            return T * 4 * S * (2 * n[lvl] + n[lvl + 1])
        vec4 = lambda l: sizes[l][0] % 4 == 0 and sizes[l][0] >= 8 and S * T >= 2  # noqa: E731  (levels of the 4-pixels-per-thread phase kernel; one frame of one stream takes the scalar one)
        # round 3: levels of >= 2^19 plane-pixels per launch with an even width take the strip form of the blur / amplify stage,
        # which recomputes the Riesz pair from the band: no per-frame pair written by the phase kernel nor read back (-8 B each)
        strips = lambda l: sizes[l][0] % 2 == 0 and n[l] * S * T >= 2000  # noqa: E731   (riesz.hip: blur_strips_min)
        if base in ("rz_phase", "rz_phase_small"):
            # per frame: band in, amp/tc/ts (+ R1/R2) out; 13 state floats R + W once per launch
            return sum((T * (16 if strips(l) else 24) + 104) * S * n[l] for l in range(nb) if vec4(l) == (base == "rz_phase"))
        if base in ("rz_seed", "rz_seed_small"):
            return sum((4 + 13 * 4) * S * n[l] for l in range(nb) if vec4(l) == (base == "rz_seed"))
        any_strips = any(strips(l) for l in range(nb))
        if base == "rz_blur_amp" and any_strips:
            return T * sum(20 * S * n[l] for l in range(nb) if strips(l))
        if base in ("rz_blur_amp", "rz_blur_amp_tiles"):
            return T * sum(28 * S * n[l] for l in range(nb) if big(l) and not strips(l))
        if base == "rz_blur_amp_small":
            return T * sum(28 * S * n[l] for l in range(nb) if not big(l) and not strips(l))
        if base == "rz_collapse" and lvl is not None:
            return T * 4 * S * (2 * n[lvl] + n[lvl + 1])
        if base == "rz_final":
            return T * S * ((4 + 3) * n[0] + 4 * n[0] + 4 * n[1])               # (ia, ib) plane in, BGR out, band + coarser result in
        return None
    if mode == "color":
        nL = n[levels]
        if base == "col_down0":
            return T * (S * ch * n[0] + 4 * P * n[1])
        if base == "col_down01":         # round 3: the first two levels in one pass, level 1 never written
            return T * (S * ch * n[0] + 4 * P * n[2])
        if base in ("pyr_down", "pyr_down_rows") and lvl is not None:
            return T * 4 * P * (n[lvl] + n[lvl + 1])
        if base == "pyr_down2" and lvl is not None:
            return T * 4 * P * (n[lvl] + n[lvl + 1] + n[lvl + 2])
        if base == "col_append":
            return T * 8 * P * nL
        if base == "col_dft":
            return 4 * P * nL * (Twin + T) + T * 4 * P * nL       # the window once per launch + one column out per frame
        if base == "col_norm":
            return T * 8 * P * nL
        if base == "pyr_up" and lvl is not None:
            return T * 4 * P * nL * (4 ** lvl + 4 ** (lvl + 1))
        nV = nL * 4 ** (levels - 1)
        if base == "col_minmax":
            return T * (S * ch * n[0] + 4 * P * nV)
        if base == "col_out":
            return T * (2 * S * ch * n[0] + 4 * P * nV)
        # round 3: k_col_out_strips makes the last TWO pyrUps itself: it reads the level-2 image of the up chain instead of V
        nU2 = nL * 4 ** max(levels - 2, 0)
        if base == "col_minmax_u2":
            return T * (S * ch * n[0] + 4 * P * nU2)
        if base == "col_out_u2":
            return T * (2 * S * ch * n[0] + 4 * P * nU2)
        return None
    return None


def batched_frame_bytes(mode, w, h, ch, levels, T, Twin):
    """Compulsory bytes per frame per stream when T frames share the launches: SURVEY.md 8d's B_alg with the
    temporal-state term divided by T (the state then crosses HBM once per call, not once per frame)."""
    sizes = level_sizes(w, h, levels)
    n = [a * b for a, b in sizes]
    io = 2.0 * ch * n[0]
    if mode == "laplace":
        return io + 16.0 * ch * sum(n[1:levels]) / T
    if mode == "riesz":
        return io + 88.0 * sum(n[0:levels - 1]) / T
    return io + (4.0 * ch * n[levels] * Twin) / min(T, 32) + 4.0 * ch * n[levels]


def reexec_under_torchrun(args):
    """`python bench.py --gpus N` from a plain shell: become N ranks (one per GPU) on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvpe(cmd[0], cmd, env)


def prime_total(lvm, pk, T, K, W):
    """priming + warm-up FRAMES (W frames of warm-up) in front of the timed region: a multiple of T"""
    need = 1 + T
    if pk["mode"] == lvm.synth.MODE_COLOR:     # the rolling window must be full before the steady state
        need = max(need, lvm.load().lvm_optimal_buffer_size(int(pk["framerate"])) + 32 + 1)
    if pk["mode"] == lvm.synth.MODE_PHASE:
        need += 1                               # the first frame passes through, the second one seeds
    return ((need + W + T - 1) // T) * T


class Runner:
    """One context + its synthetic clip(s) staged in HBM + the call schedule of the benchmark."""

    def __init__(self, lvm, torch, np, cfg_idx, small, B, ring, T, device, stream_ids, out_frames, time_shift=False):
        self.lvm, self.torch, self.np = lvm, torch, np
        ck, pk = lvm.synth.config(cfg_idx, small)
        if CLIP_NOISE is not None:
            ck = dict(ck, noise=float(CLIP_NOISE))
        self.ck, self.pk = ck, pk
        self.w, self.h, self.levels, self.ch = ck["w"], ck["h"], pk["levels"], 3
        self.B, self.ring, self.T = B, ring, T
        w, h, ch = self.w, self.h, self.ch
        self.frame_bytes = h * w * ch
        dev = torch.device("cuda", device)
        # frames are generated on the GPU in float64 with the generator's own operations (bit-identical to
        # Clip.frame, checked by tests/test_gpu_parity.py); stream s of this rank uses seed 1234 + global stream id
        self.d_in = torch.empty((ring, B, h, w, ch), dtype=torch.uint8, device=dev)
        if time_shift:      # sub-records only: B time-shifted copies of ONE clip (independent states, cheap to stage)
            clip = lvm.synth.Clip(seed=lvm.sharding.stream_seed(stream_ids[0]), **{k: v for k, v in ck.items() if k != "seed"})
            base = [clip.frame_torch(t, dev) for t in range(ring)]
            for t in range(ring):
                for s in range(B):
                    self.d_in[t, s] = base[(t + 5 * s) % ring]
        else:
            for s, sid in enumerate(stream_ids):
                clip = lvm.synth.Clip(seed=lvm.sharding.stream_seed(sid), **{k: v for k, v in ck.items() if k != "seed"})
                for t in range(ring):
                    self.d_in[t, s] = clip.frame_torch(t, dev)
        # outputs: frames 0 .. keep-1 each keep their own slot (what the verification reads back); later frames share a ring behind them
        self.keep = ((max(out_frames, 0) + T - 1) // T) * T if out_frames > ring else 0
        self.oring = ring
        self.d_out = torch.zeros((self.keep + self.oring, B, h, w, ch), dtype=torch.uint8, device=dev)
        self.ctx = lvm.Context(device, B)
        self.ctx.set_max_frames(T)
        self.cp = lvm.LvmParams(pk["mode"], pk["levels"], pk["amplification"], pk["coWavelength"], pk["coLow"], pk["coHigh"],
                                pk["chromAttenuation"], pk["framerate"], 0)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.in0 = self.d_in.data_ptr()
        self.out0 = self.d_out.data_ptr()
        self.fstride = self.frame_bytes * B
        self.fast = self.ctx.make_stepper(self.cp, w, h, ch, w * ch, self.frame_bytes, w * ch, self.frame_bytes, self.stream)
        self.fn_frames = self.ctx.lib.lvm_process_device_frames
        self.prod = (C.c_int * max(T, 1))()
        self.p_ref = C.byref(self.cp)
        self.n = 0                                # frames issued so far

    def run(self, count):
        """issue the next `count` frames in calls of at most T frames that wrap neither ring"""
        i, end = self.n, self.n + count
        w, h, ch, fb, fs = self.w, self.h, self.ch, self.frame_bytes, self.fstride
        while i < end:
            t = i % self.ring
            if i < self.keep:
                o, room = i, self.keep - i
            else:
                o = self.keep + (i - self.keep) % self.oring
                room = self.keep + self.oring - o
            nf = min(self.T, end - i, self.ring - t, room)
            if nf == 1:
                rc = self.fast(self.in0 + t * fs, self.out0 + o * fs)
            else:
                rc = self.fn_frames(self.ctx.h, self.p_ref, nf, self.in0 + t * fs, w, h, ch, w * ch, fb, fs, self.out0 + o * fs, w * ch,
                                    fb, fs, self.prod, self.stream)
            if rc != 0:
                self.ctx._check(rc)
            i += nf
        self.n = end

    def prime(self, K, W):
        """untimed: the first frame (seeds the state) and whole calls of the timed shape; ends so that the timed
        region starts on a multiple of T (ring % T == 0: timed calls are never cut by the ring).  W = warm-up FRAMES."""
        n = prime_total(self.lvm, self.pk, self.T, K, W) - W
        self.run(n)
        return n

    def out_slot(self, i):
        """where output frame i lies in d_out (None: overwritten by a later frame)"""
        if i < self.keep:
            return i
        return self.keep + (i - self.keep) % self.oring if i >= self.n - self.oring else None

    def ramp(self, seconds):
        """Untimed: keeps the GPU busy with the SAME workload on a scratch context (own state, scratch output) for `seconds`
        right before the timed region.  The measured context's frame sequence is untouched (the oracle replay stays short);
        what changes is the GPU's clock / power state: after a few milliseconds of load it is not yet the one a running
        pipeline sees (measured: 1080p Laplace 45.5 k fps after 64 warm-up frames, 53.4 k after 1600)."""
        if seconds <= 0:
            return 0
        torch = self.torch
        T = max(self.T, 1)
        n = min(T, self.ring)
        scratch = torch.empty((n, self.B, self.h, self.w, self.ch), dtype=torch.uint8, device=self.d_in.device)
        ctx = self.lvm.Context(self.d_in.device.index, self.B)
        ctx.set_max_frames(T)
        prod = (C.c_int * n)()
        w, h, ch, fb, fs = self.w, self.h, self.ch, self.frame_bytes, self.fstride
        frames = 0
        t0 = time.perf_counter()
        while True:
            for _ in range(4):
                rc = self.fn_frames(ctx.h, self.p_ref, n, self.in0, w, h, ch, w * ch, fb, fs, scratch.data_ptr(), w * ch, fb, fs, prod, self.stream)
                if rc != 0:
                    ctx._check(rc)
                frames += n
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= seconds:
                break
        # a last burst WITHOUT a trailing synchronisation: the timed region's own barrier + synchronize follows at once
        for _ in range(2):
            self.fn_frames(ctx.h, self.p_ref, n, self.in0, w, h, ch, w * ch, fb, fs, scratch.data_ptr(), w * ch, fb, fs, prod, self.stream)
            frames += n
        torch.cuda.synchronize()
        ctx.close()
        return frames

    def close(self):
        self.ctx.close()


def oracle_replay(po, np, host_frames, pk, ring, n_frames, check, threads, float_at=-1, real_reference=False):
    """The CPU oracle over frames 0 .. n_frames-1 of stream 0 (input slot i % ring); returns the oracle's u8 frames at
    the indices in `check`, its pre-quantisation float frame at index `float_at`, and the replay's wall time."""
    o = po.RefOracle() if real_reference else po.Oracle()
    P = po.make_params(**pk)
    po.lib().lvmo_set_threads(threads)
    keep = {}
    want = set(check)
    fl = None
    t0 = time.perf_counter()
    for i in range(n_frames):
        ref, produced = o.process(host_frames[i % ring], P)
        if i in want:
            keep[i] = (ref.copy(), bool(produced))
        if i == float_at and not real_reference:
            fl = o.last_float().copy()
    dt = time.perf_counter() - t0
    o.close()
    return keep, fl, dt


CLIP_NOISE = None     # --clip-noise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; a step = one call of --frames-per-call frames per stream")
    ap.add_argument("--warmup", type=int, default=5, help="untimed warm-up steps (calls) in front of the timed region")
    ap.add_argument("--ramp-ms", type=float, default=40.0,
                    help="untimed GPU load (same workload, scratch context) in front of the timed region, milliseconds")
    ap.add_argument("--mode", default="laplace", choices=list(MODES))
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU (one launch covers all)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--levels", type=int, default=0)
    ap.add_argument("--clip-noise", type=float, default=None, help="amplitude of the synthetic clip's per-channel uniform noise in levels (default 12: the headline clip; "
                    "2-3 is a camera's -- the table look-ups of the Lab modes run faster on it; the JSON's `data` says which)")
    ap.add_argument("--ring", type=int, default=64, help="distinct input frames kept in HBM (rounded up to a multiple of T)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle replay (implies --no-verify)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--verify-all-ranks", action="store_true", help="every rank checks its own stream against the oracle (tests)")
    ap.add_argument("--subrecords", action="store_true", help="run the sub-records also at N > 1 (default: N = 1 only)")
    ap.add_argument("--no-subrecords", action="store_true")
    ap.add_argument("--frames-per-call", type=int, default=32, help="consecutive frames of the stream(s) handed to one lvm_process_device_frames call (1 = per-frame calls)")
    ap.add_argument("--pipeline", type=int, default=0, help="cross-frame pipeline depth of lvm_process_device (0 or 1)")
    ap.add_argument("--profile-steps", type=int, default=-1, help="frames of the per-kernel timing pass (default: two calls of T frames; 0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (smoke-testing the N > 1 path)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--no-clock-probe", action="store_true", help="no shader-clock probe beside the timed region (a profiler that serialises kernels would "
                    "make the timed calls wait for the probe's time-out: tools/pmc.sh passes this)")
    args = ap.parse_args()
    global CLIP_NOISE
    CLIP_NOISE = args.clip_noise
    global RAMP_SECONDS
    RAMP_SECONDS = args.ramp_ms * 1e-3

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        reexec_under_torchrun(args)

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
    host_binding = bind_to_gpu_numa_node(torch, local_rank)
    dist = None
    # One rank per GPU under the launcher (torch.distributed.run sets WORLD_SIZE, also for --nproc-per-node 1): the process
    # group exists at EVERY world size then, so that the barrier and the MAX-reduce of the timed region run through RCCL
    # (backend "nccl" on ROCm) at N = 1 exactly as they will at N = 8.  A plain `python bench.py` has no rendezvous and none.
    if world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    red_dev = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else torch.device("cpu")
    dist_info = {"initialized": dist is not None, "backend": (dist.get_backend() if dist is not None else None),
                 "world_size": (dist.get_world_size() if dist is not None else 1)}
    if dist is not None:
        # first contact: a device-tensor all-reduce (SUM of ones = the number of ranks RCCL really connected) + barrier
        one = torch.ones(1, dtype=torch.int32, device=red_dev)
        dist.all_reduce(one)
        dist.barrier()
        dist_info["ranks_seen_by_allreduce"] = int(one.item())

    lvm = importlib.import_module("live-video-magnification_amd")
    cfg_idx = MODES[args.mode]
    small = None
    ck0, pk0 = lvm.synth.config(cfg_idx)
    if args.width or args.height or args.levels:
        small = (args.width or ck0["w"], args.height or ck0["h"], args.levels or pk0["levels"])
    B = args.streams
    K, W = args.steps, args.warmup
    T = max(1, args.frames_per_call)
    ring = ((max(args.ring, T) + T - 1) // T) * T
    T = min(T, ring)
    verify = not (args.no_verify or args.no_cpu_baseline) and (args.verify_all_ranks or (rank == 0 and world == 1))
    do_sub = (world == 1 or args.subrecords) and not args.no_subrecords

    ids = lvm.sharding.stream_ids(rank, world, B)
    # K steps = K calls of T frames; W warm-up steps likewise.  Outputs of every frame up to the end of the verified span keep their own
    # slot in HBM (the verification reads the timed region's OWN output back); later frames share a ring.
    Kf, Wf = K * T, W * T                                            # frames
    pk_s = lvm.synth.config(cfg_idx, small)[1]
    base_expected = prime_total(lvm, pk_s, T, K, Wf) + T             # priming + warm-up + the float-probe call
    v_calls = min(K, 12) if T > 1 else min(K, 384)                   # timed calls the oracle replays through (bounds the CPU work)
    v_end = base_expected + v_calls * T
    R = Runner(lvm, torch, np, cfg_idx, small, B, ring, T, local_rank, ids, v_end if verify else ring)
    w, h, levels, ch, pk = R.w, R.h, R.levels, R.ch, R.pk
    R.ctx.set_pipeline(args.pipeline)
    stream = R.stream

    primed = R.prime(K, Wf)
    R.run(Wf)
    # ---- float probe: one more call with the pre-quantisation frame kept (stream 0); in front of the timed region so that the oracle
    # replay can stop after the verified span ----
    probe_n = 0
    float_gpu = None
    probe_first = -1
    if verify:
        R.ctx.flush(stream)
        pd = args.pipeline
        R.ctx.set_pipeline(0)
        R.ctx.keep_float(True)
        probe_n = T
        probe_first = R.n                # the kernels keep the float frame of the FIRST frame of a batch (stream 0)
        if args.mode == "color":
            probe_first += 32 * ((probe_n - 1) // 32)      # the colour mode cuts a call into chunks of <= 32 frames
        R.run(probe_n)
        torch.cuda.synchronize()
        float_gpu = R.ctx.read_float((h, w, ch))
        R.ctx.keep_float(False)
        R.ctx.set_pipeline(pd)
    base = R.n
    assert not verify or base == base_expected, (base, base_expected)
    ramp_frames = R.ramp(args.ramp_ms * 1e-3)
    # ---- the timed region: K calls ----
    dt = lvm.sharding.timed_steps(lambda _: R.run(Kf), 1, dist, torch.cuda.synchronize, red_dev, finish=lambda: R.ctx.flush(stream))
    host_enqueue = getattr(lvm.sharding.timed_steps, "host_seconds", 0.0)
    fps = lvm.sharding.aggregate_fps(world, B, Kf, dt)
    n_verify = min(R.n, v_end)
    # ---- the shader clock under this workload: the SAME K calls once more, right behind the timed region, with a one-lane kernel on the auxiliary
    # stream reading s_memtime / s_memrealtime at both ends (lvm_debug_clock_probe_*).  Not inside the timed region: the probe's wave takes a slot that
    # the persistent / one-round launches count on -- measured on this part: Laplace -1..2 %, Riesz 0, Color -6..10 % (profiles/r06_clock_probe_cost.txt).
    clock = {"mhz": None}
    if not args.no_clock_probe:
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            R.ctx.clock_probe_start(2.0)
            R.run(Kf)
            R.ctx.flush(stream)
            if stream == 0:          # handle 0 (torch's default stream) makes the library use the context's own stream
                R.ctx.synchronize()
            else:
                torch.cuda.current_stream().synchronize()
            clock["mhz"], clock["seconds"] = R.ctx.clock_probe_stop()
            torch.cuda.synchronize()
            clock["fps_with_probe"] = round(B * Kf / (time.perf_counter() - t0), 2)
        except Exception as e:      # a measurement aid must never take the headline down
            clock["error"] = str(e)[:120]

    # ---- verification against the CPU oracle (also the cpu_baseline sample) ----
    cpu = None
    verified = None
    vinfo = None
    if verify:
        from oracle import pyoracle as po
        nthreads = max(1, min(16, os.cpu_count() or 1))   # beyond ~16 threads fork/join over small levels dominates
        host = R.d_in[:, 0].cpu().numpy()                 # stream 0 of this rank
        timed_idx = sorted(set(int(round(x)) for x in np.linspace(base, n_verify - 1, 12)))
        check = [1, primed - 1] + timed_idx
        keep, fl_ref, cdt = oracle_replay(po, np, host, pk, ring, n_verify, check, nthreads, float_at=probe_first)
        worst_du, worst_frac, ok = 0, 1.0, True
        n_cmp = 0
        for i in check:
            ref, produced = keep[i]
            if not produced:
                continue
            got = R.d_out[i, 0].cpu().numpy()             # (i < R.keep: every verified frame has its own slot)
            du = np.abs(ref.astype(np.int16) - got.astype(np.int16))
            worst_du = max(worst_du, int(du.max()))
            worst_frac = min(worst_frac, float((du == 0).mean()))
            n_cmp += 1
        rel = float(np.abs(fl_ref - float_gpu).max() / max(float(np.abs(fl_ref).max()), 1e-30))
        ok = n_cmp >= 8 and worst_du <= 1 and worst_frac >= 0.999 and rel <= 1e-4 and bool(np.isfinite(float_gpu).all())
        verified = bool(ok)
        vinfo = {"frames_compared": n_cmp, "timed_frames_compared": len([i for i in check if base <= i < base + Kf]),
                 "timed_calls_covered": v_calls, "u8_max_diff": worst_du, "u8_identical_min": round(worst_frac, 6), "float_rel_err_probe": rel,
                 "bars": "u8 <= 1 LSB, >= 99.9 % identical; float <= 1e-4 of max|ref|", "oracle_frames_replayed": n_verify}
        if not np.isfinite(float_gpu).all() or not np.isfinite(fl_ref).all():      # say where: a non-finite probe must be diagnosable from the record
            bad = np.argwhere(~np.isfinite(float_gpu))
            vinfo["float_nonfinite"] = {"gpu_count": int(len(bad)), "gpu_first": bad[:6].tolist(), "gpu_last": bad[-3:].tolist(),
                                        "oracle_count": int((~np.isfinite(fl_ref)).sum()), "probe_frames": int(probe_n), "probe_first": int(probe_first)}
        ref_check = None
        if po.RefOracle.available():
            # the REAL reference stage (oracle/_ref/libref_magnify.so, built where OpenCV 4 exists): reported next to
            # `verified`.  Since round 3 the library computes OpenCV's interpolated forward Lab; whether its restated table
            # equals the one of this OpenCV build is reported too (a differing table can be installed with lvm_set_lab_lut)
            nref = min(n_verify, base + 64)
            rkeep, _, rdt = oracle_replay(po, np, host, pk, ring, nref, [i for i in check if i < nref], nthreads, real_reference=True)
            dmax, fmin = 0, 1.0
            for i, (ref, produced) in rkeep.items():
                if produced:
                    du = np.abs(ref.astype(np.int16) - R.d_out[i, 0].cpu().numpy().astype(np.int16))
                    dmax, fmin = max(dmax, int(du.max())), min(fmin, float((du == 0).mean()))
            ref_check = {"frames": len(rkeep), "u8_max_diff": dmax, "u8_identical_min": round(fmin, 6), "fps": round(nref / rdt, 3)}
            rt = po.RefOracle.recover_lab_lut()
            ref_check["lab_lut_equals_opencv"] = None if rt is None else bool(np.array_equal(rt, R.ctx.lab_lut()))
        _, _, cdt1 = oracle_replay(po, np, host, pk, ring, 8, [], 1)     # the same clip's first 8 frames on ONE thread
        cpu = {"value": round(n_verify / cdt, 3), "unit": "frames/s", "cores": nthreads, "kind": "port",
               "host_cores": os.cpu_count(), "threads_used": nthreads,
               "single_thread": {"value": round(8 / cdt1, 3), "unit": "frames/s", "cores": 1, "sample": "first 8 frames of the same clip"},
               "sample": "%d frames of the same %dx%d L%d %s clip (the verification replay), CPU oracle = restatement of the "
                         "reference, OpenMP over rows, %d threads of the host's %s cores" % (n_verify, w, h, levels, args.mode, nthreads, os.cpu_count()),
               "reference_probe": probe_reference(), "real_reference_check": ref_check}
        if ref_check:
            cpu.update({"kind": "reference", "value": ref_check["fps"], "cores": 1,
                        "sample": "the reference's own MagnificationProcessor (oracle/_ref/libref_magnify.so) on the first %d frames" % nref,
                        "port_value": round(n_verify / cdt, 3)})
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        nthreads = max(1, min(16, os.cpu_count() or 1))
        host = R.d_in[:, 0].cpu().numpy()
        _, _, cdt = oracle_replay(po, np, host, pk, ring, 64, [], nthreads)
        _, _, cdt1 = oracle_replay(po, np, host, pk, ring, 8, [], 1)
        cpu = {"value": round(64 / cdt, 3), "unit": "frames/s", "cores": nthreads, "kind": "port",
               "host_cores": os.cpu_count(), "threads_used": nthreads,
               "single_thread": {"value": round(8 / cdt1, 3), "unit": "frames/s", "cores": 1, "sample": "first 8 frames of the same clip"},
               "sample": "64 frames of the same clip, CPU oracle", "reference_probe": probe_reference()}

    # ---- per-kernel timing pass (HIP events on the launch stream) -> roofline ----
    roofline = None
    kernels = {}
    Twin = lvm.load().lvm_optimal_buffer_size(int(pk["framerate"]))
    if args.profile_steps < 0:
        args.profile_steps = 2 * T if T > 1 else 60        # whole calls only: every launch then covers exactly T frames
    if rank == 0 and args.profile_steps > 0:
        kernels, roofline = kernel_pass(lvm, torch, R, args.mode, args.profile_steps, args.ramp_ms * 1e-3)
    b_alg = lvm.load().lvm_algorithmic_bytes(pk["mode"], w, h, ch, levels, pk["framerate"])
    frame_frac = b_alg * (fps / world / B) / (HBM_PEAK_GBS * 1e9)
    b_bat = batched_frame_bytes(args.mode, w, h, ch, levels, T, Twin)
    if roofline is not None:
        roofline["frame_frac"] = round(frame_frac, 5)      # SURVEY.md 8d's figure for the WHOLE path next to the dominant kernel's own
        roofline["frame_frac_definition"] = "B_alg x frames/s / peak: compulsory bytes of the reference algorithm per frame (no credit for intermediates)"
        roofline["frame_alg_bytes"] = b_alg

    # ---- gather per-rank facts (tests check value == world * B * K / max dt and every rank's verification) ----
    props = torch.cuda.get_device_properties(local_rank)
    rank_facts = [{"rank": rank, "verified": verified, "verification": vinfo, "stream_ids": ids, "device": "cuda:%d" % local_rank, "host_binding": host_binding,
                   "device_name": props.name, "device_uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_facts[0])
        rank_facts = gathered
    R.close()
    del R
    torch.cuda.empty_cache()

    # ---- the timed region's own shape once more WITHOUT the clock ramp (a GPU a few milliseconds out of idle) ----
    value_cold = None
    if not args.no_subrecords:
        Rc = Runner(lvm, torch, np, cfg_idx, small, B, ring, T, local_rank, ids, ring)
        Rc.ctx.set_pipeline(args.pipeline)
        dtc = timed_run(lvm, torch, Rc, Kf, Wf, dist, red_dev, ramp=0.0)
        value_cold = {"value": round(lvm.sharding.aggregate_fps(world, B, Kf, dtc), 2), "unit": "frames/s", "steps": K, "warmup": W, "ramp_ms": 0.0,
                      "note": "same call shape as `value`, priming + warm-up only: the clocks have not ramped"}
        Rc.close()
        del Rc
        torch.cuda.empty_cache()

    # ---- sub-records: the other schedules / surfaces SURVEY.md 8d asks for, measured in the same run ----
    sub = {}
    if do_sub:
        sub = sub_records(lvm, torch, np, args, cfg_idx, small, local_rank, rank, world, dist, red_dev)
    elif world > 1:
        sub = {"cfg4_riesz_4k": config_record(lvm, torch, np, 4, local_rank, rank, world, dist, red_dev, K=64, W=16, T=16, ring=16, extras=False),
               "host_fed": host_fed_record(lvm, torch, np, cfg_idx, small, local_rank, rank, world, dist, red_dev)}

    if rank == 0:
        out = {
            "metric": "magnified frames/sec at 1080p, Laplace-motion 6 levels; % HBM roofline" if args.mode == "laplace"
                      else "magnified frames/sec (%s)" % args.mode,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * dt / K, 5), "frames_per_step": T, "timed_frames_per_stream": Kf, "us_per_frame": round(1e6 * dt / Kf, 4),
            "step_definition": "one lvm_process_device_frames call of frames_per_step consecutive frames per stream (one pass of the hot path over one batch)",
            "clock_mhz": (round(clock["mhz"], 1) if clock.get("mhz") else None),
            "clock": {"mhz": (round(clock["mhz"], 1) if clock.get("mhz") else None), "seconds_covered": clock.get("seconds"), "error": clock.get("error"),
                      "frames_per_s_of_the_probed_pass": clock.get("fps_with_probe"),
                      "how": "average shader clock over a SECOND pass of the same K steps right behind the timed region: s_memtime / s_memrealtime x 100 MHz, read by a "
                             "one-lane kernel on a second stream at both ends of the pass (lvm_debug_clock_probe_*); outside the timed region because the probe's wave "
                             "costs the timed kernels 0-10 % (this rank only)"},
            "value_cold": value_cold,
            "host_enqueue_ms_per_step": round(1e3 * host_enqueue / K, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if CLIP_NOISE is None else "synthetic (clip noise +-%g levels instead of +-12)" % CLIP_NOISE,
            "config": {"workload": "%s %dx%d, %d levels, %d stream(s)/GPU, device-resident u8 BGR in/out" %
                                   (args.mode, w, h, levels, B),
                       "baseline_config": cfg_idx, "streams_per_gpu": B, "ring_frames": ring,
                       "frames_per_call": T, "priming_frames": primed, "ramp_ms": args.ramp_ms, "ramp_frames_scratch_context": ramp_frames, "pipeline_depth": args.pipeline},
            "verified": verified, "verification": vinfo,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "frame_alg_bytes": b_alg, "frame_roofline_frac": round(frame_frac, 5),
            "frame_batched_alg_bytes": round(b_bat), "frame_batched_roofline_frac": round(b_bat * (fps / world / B) / (HBM_PEAK_GBS * 1e9), 5),
            "timed_seconds_max_over_ranks": dt,
            "dist": dist_info,
            "ranks": rank_facts,
            "kernels": kernels,
        }
        out.update(sub)
        if "per_frame" in sub:      # what MagnificationProcessor::process itself runs (MagnificationProcessor.cpp:17-67): T = 1, one stream
            pf = sub["per_frame"]
            out["process_schedule"] = {"value": pf.get("value"), "unit": "frames/s", "us_per_frame": pf.get("us_per_frame"), "frames_per_call": 1, "streams": 1,
                                       "frame_roofline_frac": pf.get("frame_roofline_frac"), "launches_per_frame": pf.get("launches_per_frame"),
                                       "note": "one lvm_process_device call per frame, device-resident: the reference's per-frame process() schedule; the headline is the batched surface"}
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


def kernel_pass(lvm, torch, R, mode, profile_steps, ramp_s):
    """Per-kernel HIP-event pass over `profile_steps` frames of R's schedule (whole calls) -> (kernel table, roofline of the
    dominant kernel).  `achieved` = the kernel's algorithmic bytes per launch (kernel_alg_bytes) / its average duration with ONLY
    its own launches bracketed; `traffic` / `rocprof_avg_us_stored` are STORED values of the rocprofv3 passes of the same command
    (profiles/<round>_pmc_traffic_<mode>.json, tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes, calibrated with
    tools/hbm_counter_calib.hip) -- a bench run cannot wrap itself in rocprofv3."""
    w, h, ch, levels, B, T, pk = R.w, R.h, R.ch, R.levels, R.B, R.T, R.pk
    stream = R.stream
    Twin = lvm.load().lvm_optimal_buffer_size(int(pk["framerate"]))
    kernels = {}
    roofline = None
    R.ctx.flush(stream)
    R.ctx.set_pipeline(0)           # per-kernel event timing uses the plain schedule
    R.run((-R.n) % T)               # whole calls
    R.run(2 * T if T > 1 else 60)   # un-timed: an oracle replay may have left the GPU idle for seconds, let the clocks settle
    R.ramp(ramp_s)
    R.ctx.profile(True)
    R.run(profile_steps)
    torch.cuda.synchronize()
    prof = R.ctx.profile_collect()
    R.ctx.profile(False)
    if not prof:
        return kernels, roofline
    tot = sum(v[0] for v in prof.values()) or 1.0
    T_launch = min(T, 32) if mode == "color" else T     # the colour mode cuts a call into chunks of <= 32 frames
    for name, (ms, cnt) in prof.items():
        avg_us = 1e3 * ms / max(cnt, 1)
        ab = kernel_alg_bytes(mode, name, w, h, ch, levels, B, T_launch, Twin)
        kernels[name] = {"avg_us": round(avg_us, 3), "launches": cnt, "share": round(ms / tot, 4),
                         "alg_bytes": ab, "gbs": (round(ab / (avg_us * 1e-6) / 1e9, 1) if ab else None)}
    dom = max(prof.items(), key=lambda kv: kv[1][0])[0]
    k = kernels[dom]
    # the dominant kernel once more with ONLY its launches bracketed: its neighbours then run back to back as in the
    # timed region (and under rocprofv3); with an event gap on both sides the VALU-bound kernels run ~10 % faster
    # (the clock recovers in the gaps), which is not the duration the frame pays for
    R.ctx.profile_only(dom)
    R.ctx.profile(True)
    R.run(profile_steps)
    torch.cuda.synchronize()
    solo = R.ctx.profile_collect().get(dom)
    R.ctx.profile(False)
    R.ctx.profile_only(None)
    k = dict(k)
    k["avg_us_all_bracketed"] = k["avg_us"]
    if solo and solo[1]:
        k["avg_us"] = round(1e3 * solo[0] / solo[1], 3)
        k["gbs"] = round(k["alg_bytes"] / (k["avg_us"] * 1e-6) / 1e9, 1) if k["alg_bytes"] else None
    traffic = None
    rocprof_avg = None
    traffic_src = "profiles/%s_pmc_traffic_%s%s.json" % (PROFILE_ROUND, mode, "" if (w, h) == (1920, 1080) else "_%dx%d" % (w, h))
    try:
        pm = json.load(open(os.path.join(ROOT, traffic_src)))
        if pm.get("key") == "%s|%dx%d|L%d|B%d|T%d" % (mode, w, h, levels, B, T):
            traffic = pm["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
            rocprof_avg = pm["kernels"].get(dom, {}).get("rocprof_avg_us")
            if traffic is not None and k["alg_bytes"] and traffic < 0.97 * k["alg_bytes"]:
                traffic = None          # cache hits can hide re-reads, never compulsory bytes: an uncalibrated counter
    except Exception:
        traffic = None
    if k["gbs"]:
        roofline = {"bound": "hbm", "kernel": dom, "achieved": k["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(k["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_us": k["avg_us"], "avg_us_all_kernels_bracketed": k["avg_us_all_bracketed"],
                    "timing": "HIP events on the launch stream around this kernel only (neighbours back to back)",
                    "alg_bytes_per_launch": k["alg_bytes"], "frames_per_launch": T_launch,
                    "traffic_source": ("stored: %s (rocprofv3 --pmc passes of this command, calibrated, tools/pmc_traffic.py)" % traffic_src)
                                      if traffic is not None else None}
        if dom in ("lap_down0_lut", "lab_lut"):
            roofline["limiter"] = ("OpenCV's forward Lab table: 8 random LDS reads + one 16-byte gather from L2 + ~80 VALU "
                                   "operations per pixel (lab_lut.h); HBM idles")
        if rocprof_avg is not None:
            roofline["rocprof_avg_us_stored"] = rocprof_avg
    return kernels, roofline


def probe_reference():
    """SURVEY.md 8c / BASELINE.md step 1: is a real OpenCV (hence the real reference) available on this box?"""
    found = {"cv2": False, "opencv_cmake_or_pc": False}
    try:
        import cv2  # noqa: F401
        found["cv2"] = getattr(cv2, "__version__", True)
    except Exception:
        pass
    for d in ("/usr/lib/x86_64-linux-gnu/cmake/opencv4", "/usr/local/lib/cmake/opencv4", "/usr/lib/cmake/opencv4",
              "/usr/lib/x86_64-linux-gnu/pkgconfig/opencv4.pc", "/usr/local/lib/pkgconfig/opencv4.pc", "/usr/include/opencv4"):
        if os.path.exists(d):
            found["opencv_cmake_or_pc"] = d
    found["oracle_ref_built"] = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_magnify.so"))
    found["note"] = ("real reference stage available" if found["oracle_ref_built"] else
                     "no OpenCV 4 on the build host: the reference stage is unbuildable, cpu_baseline is the restatement")
    return found


RAMP_SECONDS = 0.04      # set from --ramp-ms in main()


def timed_run(lvm, torch, R, K, W, dist, red_dev, ramp=None):
    """K timed FRAMES after W warm-up frames, in calls of R.T (the sub-records count frames)"""
    R.prime(K, W)
    R.run(W)
    R.ramp(RAMP_SECONDS if ramp is None else ramp)
    dt = lvm.sharding.timed_steps(lambda i: R.run(K), 1, dist, torch.cuda.synchronize, red_dev, finish=lambda: R.ctx.flush(R.stream))
    return dt


def per_frame_records(lvm, torch, np, cfg_idx, small, local_rank, ids, world, dist, red_dev, streams=(1, 4), K=None):
    """The schedule MagnificationProcessor::process really runs (MagnificationProcessor.cpp:17-67): ONE frame per call (T = 1), every
    temporal state word read and written every frame; B independent streams per launch (B in {1, 4, 16}).
    Device-resident frames.  Returns {"B1": {...}, "B4": {...}, ...}."""
    out = {}
    for Bn in streams:
        R = Runner(lvm, torch, np, cfg_idx, small, Bn, 32 if Bn == 1 else 16, 1, local_rank, ids, 32 if Bn == 1 else 16, time_shift=Bn > 1)
        big = R.w * R.h > 1920 * 1080
        Kq = K or ((300 if Bn == 1 else 120) if not big else 48)
        dt = timed_run(lvm, torch, R, Kq, 16, dist, red_dev)
        v = world * Bn * Kq / dt
        b_alg = lvm.load().lvm_algorithmic_bytes(R.pk["mode"], R.w, R.h, R.ch, R.levels, R.pk["framerate"])
        out["B%d" % Bn] = {"schedule": "T = 1: one lvm_process_device call per frame, device-resident", "value": round(v, 2), "unit": "frames/s",
                           "streams": Bn, "frames_per_call": 1, "steps": Kq, "us_per_frame": round(1e6 * dt / (Kq * Bn), 3),
                           "host_enqueue_us_per_frame": round(1e6 * lvm.sharding.timed_steps.host_seconds / (Kq * Bn), 2),
                           "frame_alg_bytes": b_alg, "frame_roofline_frac": round(b_alg * v / world / (HBM_PEAK_GBS * 1e9), 5)}
        R.close()
        del R
        torch.cuda.empty_cache()
    return out


def e2e_host_record(lvm, np, ctx, cp_ref, host, kinds=("pageable", "pinned"), Ke=150, warm=20):
    """host u8 frame in -> host u8 frame out through lvm_process (the drop-in surface), one frame in flight."""
    lib = lvm.load()
    _, h, w, ch = host.shape
    fb = h * w * ch
    e2e = {}
    for kind in kinds:
        staged = kind == "pageable_staged_through_pinned"   # what a pinned staging buffer INSIDE lvm_process would cost: two memcpys
        if kind != "pageable":
            pin, pout = C.c_void_p(), C.c_void_p()
            if lib.lvm_host_alloc(fb * 8, C.byref(pin)) != 0 or lib.lvm_host_alloc(fb, C.byref(pout)) != 0:
                continue
            src = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(8, h, w, ch))
            dst = np.ctypeslib.as_array(C.cast(pout, C.POINTER(C.c_uint8)), shape=(h, w, ch))
            src[:] = host[:8]
            if kind == "pageable_in_pinned_out":      # the one-line swap as it is: FramePool frames pageable, the shim's output from its pinned pool
                src = host[:8].copy()
            if staged:
                pg_src, pg_dst = host[:8].copy(), np.empty((h, w, ch), np.uint8)
        else:
            src = host[:8].copy()
            dst = np.empty((h, w, ch), np.uint8)
        produced = C.c_int(0)
        fn = lib.lvm_process
        ptrs = [src[i].ctypes.data for i in range(8)]
        po_ = dst.ctypes.data
        ctx.reset()
        for i in range(warm):           # (colour mode: until the rolling window is full)
            fn(ctx.h, cp_ref, ptrs[i % 8], w, h, ch, w * ch, po_, w * ch, C.byref(produced))
        t0 = time.perf_counter()
        for i in range(Ke):
            if staged:
                np.copyto(src[i % 8], pg_src[i % 8])
            rc = fn(ctx.h, cp_ref, ptrs[i % 8], w, h, ch, w * ch, po_, w * ch, C.byref(produced))
            if rc != 0:
                ctx._check(rc)
            if staged:
                np.copyto(pg_dst, dst)
        de = time.perf_counter() - t0
        e2e[kind] = {"value": round(Ke / de, 2), "unit": "frames/s", "us_per_frame": round(1e6 * de / Ke, 1),
                     "pcie_bytes_per_frame": 2 * fb, "pcie_gbs": round(2 * fb * Ke / de / 1e9, 2)}
        if kind != "pageable":
            del src, dst
            lib.lvm_host_free(pin); lib.lvm_host_free(pout)
    return dict(e2e, surface="lvm_process: host u8 in -> host u8 out, synchronous (upload, kernels and download of ONE frame overlap "
                             "row-chunk by row-chunk; replaces MagnificationProcessor::process, MagnificationProcessor.cpp:17-67)")


def export_host_record(lvm, np, local_rank, cp_ref, host, Te=32, Kx=6):
    """lvm_export_frames: Te page-locked host frames in, Te side-by-side canvases out per call (the loop body of Exporter::run,
    Exporter.cpp:216-259: runChainOnce + compose); uploads, kernels and downloads of consecutive sub-batches overlap."""
    try:
        lib = lvm.load()
        _, h, w, ch = host.shape
        fb = h * w * ch
        cpre = lvm.LvmPreprocessParams(1, 0, 0.0, 0.0, 1.0, 1.0, 0)
        cw, chh = C.c_int(0), C.c_int(0)
        lib.lvm_export_geometry(C.byref(cpre), 1, w, h, ch, C.byref(cw), C.byref(chh))
        cb = cw.value * chh.value * 3
        pin, pout = C.c_void_p(), C.c_void_p()
        if not (cb > 0 and lib.lvm_host_alloc(fb * Te, C.byref(pin)) == 0 and lib.lvm_host_alloc(cb * Te, C.byref(pout)) == 0):
            return {"error": "page-locked allocation failed"}
        src = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(Te, h, w, ch))
        for i in range(Te):
            src[i] = host[i % host.shape[0]]
        vp = C.c_void_p
        pi = (vp * Te)(*[pin.value + i * fb for i in range(Te)])
        pc = (vp * Te)(*[pout.value + i * cb for i in range(Te)])
        prod = (C.c_int * Te)()
        ex = lvm.Context(local_rank, 1)
        ex.set_max_frames(Te)
        for _ in range(2):
            ex._check(lib.lvm_export_frames(ex.h, C.byref(cpre), cp_ref, 1, Te, pi, w, h, ch, w * ch, pc, cw.value * 3, prod))
        t0 = time.perf_counter()
        for _ in range(Kx):
            ex._check(lib.lvm_export_frames(ex.h, C.byref(cpre), cp_ref, 1, Te, pi, w, h, ch, w * ch, pc, cw.value * 3, prod))
        dx = time.perf_counter() - t0
        ex.close()
        # the same loop body with the canvases JPEG-encoded on the device (ExportFormat::AviMjpg, Exporter.cpp:107-117 + :259): only the
        # compressed frames come down (into the canvas slots), cv::VideoWriter's software codec is out of the loop
        mj = None
        try:
            q = 85
            ej = lvm.Context(local_rank, 1)
            ej.set_max_frames(Te)
            offs = (C.c_size_t * (Te + 1))()
            for _ in range(2):
                ej._check(lib.lvm_export_frames_mjpeg(ej.h, C.byref(cpre), cp_ref, 1, Te, pi, w, h, ch, w * ch, q, pout, cb * Te, offs, prod))
            t0 = time.perf_counter()
            for _ in range(Kx):
                ej._check(lib.lvm_export_frames_mjpeg(ej.h, C.byref(cpre), cp_ref, 1, Te, pi, w, h, ch, w * ch, q, pout, cb * Te, offs, prod))
            dj = time.perf_counter() - t0
            ej.close()
            jb = offs[Te] / Te
            mj = {"value": round(Kx * Te / dj, 2), "unit": "frames/s", "us_per_frame": round(1e6 * dj / (Kx * Te), 1), "quality": q,
                  "jpeg_bytes_per_frame": int(jb), "bits_per_canvas_pixel": round(8.0 * jb / (cb / 3), 3), "pcie_bytes_per_frame": int(fb + jb),
                  "surface": "lvm_export_frames_mjpeg: the same loop body, canvases -> baseline JPEG (4:2:0, restart interval per MCU row) on the "
                             "device, byte-identical to oracle/mjpeg_oracle.py and decoded by libjpeg in tests/test_mjpeg.py; "
                             "host/HipMjpegWriter.hpp is the AVI container"}
        except Exception as e:
            mj = {"error": str(e)[:200]}
        del src
        lib.lvm_host_free(pin); lib.lvm_host_free(pout)
        return {"value": round(Kx * Te / dx, 2), "unit": "frames/s", "frames_per_call": Te, "us_per_frame": round(1e6 * dx / (Kx * Te), 1),
                "pcie_bytes_per_frame": fb + cb, "pcie_gbs": round((fb + cb) * Kx * Te / dx / 1e9, 2),
                "surface": "lvm_export_frames: pinned host frames in -> side-by-side canvases (original | processed) out; replaces the loop "
                           "body of Exporter::run (Exporter.cpp:216-259: runChainOnce + compose), host/HipExportRunner.hpp is the loop",
                "mjpeg": mj}
    except Exception as e:      # a sub-record must never take the headline line down
        return {"error": str(e)[:200]}


def bind_to_gpu_numa_node(torch, local_rank):
    """One rank per GPU: pin this process (and the threads it starts) to the CPUs of the NUMA node its GPU hangs off, so that
    page-locked frame buffers are allocated there and the PCIe DMA does not cross the socket interconnect (the
    host-fed aggregate is limited by host feeding / PCIe).  Returns what was done, for the JSON line."""
    info = {"bound": False}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        info["pci"] = bdf
        node = int(open(base + "/numa_node").read().strip())
        info["numa_node"] = node
        cpulist = open(base + "/local_cpulist").read().strip()
        cpus = set()
        for part in cpulist.split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        if cpus and os.environ.get("LVM_BENCH_BIND", "1") != "0":
            os.sched_setaffinity(0, cpus)
            info.update(bound=True, cpus=len(cpus), cpulist=cpulist)
    except Exception as e:      # no sysfs in the container, no such attribute: run unbound and say so
        info["note"] = "not bound: %s" % str(e)[:120]
    return info


def host_fed_record(lvm, torch, np, cfg_idx, small, local_rank, rank, world, dist, red_dev):
    """N > 1: the host-fed surfaces on EVERY rank at the same time (north_star's ">= 6x at 8 GPUs" is limited by host feeding and
    PCIe, SURVEY.md 8e -- device-resident streams cannot show that): each rank, bound to its GPU's NUMA node, runs lvm_process on
    page-locked frames and lvm_export_frames on 32-frame batches between barriers; rank 0 reports the aggregate and the per-rank
    PCIe rates."""
    ids = lvm.sharding.stream_ids(rank, world, 1)
    R = Runner(lvm, torch, np, cfg_idx, small, 1, 32, 1, local_rank, ids, 32)
    host = R.d_in[:, 0].cpu().numpy()
    if dist is not None:
        dist.barrier()
    e2e = e2e_host_record(lvm, np, R.ctx, R.p_ref, host, kinds=("pinned",), Ke=120)
    if dist is not None:
        dist.barrier()
    ex = export_host_record(lvm, np, local_rank, R.p_ref, host, Kx=4)
    R.close()
    mine = {"rank": rank, "e2e_fps": e2e.get("pinned", {}).get("value"), "e2e_pcie_gbs": e2e.get("pinned", {}).get("pcie_gbs"),
            "export_fps": ex.get("value"), "export_pcie_gbs": ex.get("pcie_gbs")}
    allr = [mine]
    if dist is not None:
        allr = [None] * world
        dist.all_gather_object(allr, mine)
    tot = lambda k: round(sum((r.get(k) or 0.0) for r in allr), 2)      # noqa: E731
    # every rank moves the same number of frames between the same two barriers: all frames / the SLOWEST rank's time = N x min(rate)
    agg = lambda k: round(len(allr) * min((r.get(k) or 0.0) for r in allr), 2)      # noqa: E731
    return {"note": "every rank runs the host -> host surfaces concurrently (own context, own GPU, own page-locked buffers); value = all ranks' frames / the "
                    "slowest rank's time",
            "e2e_host_pinned": {"value": agg("e2e_fps"), "unit": "frames/s", "pcie_gbs_total": tot("e2e_pcie_gbs")},
            "export_host": {"value": agg("export_fps"), "unit": "frames/s", "pcie_gbs_total": tot("export_pcie_gbs")},
            "per_rank": allr}


def config_record(lvm, torch, np, cfg_idx, local_rank, rank, world, dist, red_dev, K=64, W=32, T=32, ring=64, extras=True, clip_tag=""):
    """One of BASELINE.json's configs measured in the same run as the headline, so that the driver's record carries it:
    configs[2] / configs[3] (Riesz / Color at 1920x1080, 6 levels) and configs[4] (Riesz 3840x2160, 8 levels, one stream per GPU;
    at every N).  K timed frames in calls of T on every rank (MAX over ranks); on rank 0 also: the timed region's own output
    against the CPU oracle (>= 8 frames, u8 bars) -> `verified`, the replay's rate -> `cpu_baseline`, the per-kernel HIP-event pass
    -> `roofline` (dominant kernel; `frame_frac` = SURVEY 8d's B_alg x fps / peak), and -- `extras` -- the per-frame schedule
    (T = 1, B = 1 / 4) and the host -> host surface (lvm_process)."""
    ids = lvm.sharding.stream_ids(rank, world, 1)
    pk = lvm.synth.config(cfg_idx)[1]
    total = prime_total(lvm, pk, T, K, W) + K
    R = Runner(lvm, torch, np, cfg_idx, None, 1, ring, T, local_rank, ids, total if rank == 0 else ring)
    mode = {lvm.synth.MODE_LAPLACE: "laplace", lvm.synth.MODE_PHASE: "riesz", lvm.synth.MODE_COLOR: "color"}[R.pk["mode"]]
    R.prime(K, W)
    R.run(W)
    base = R.n
    R.ramp(RAMP_SECONDS)
    dt = lvm.sharding.timed_steps(lambda i: R.run(K), 1, dist, torch.cuda.synchronize, red_dev, finish=lambda: R.ctx.flush(R.stream))
    fps = world * K / dt
    b_alg = lvm.load().lvm_algorithmic_bytes(R.pk["mode"], R.w, R.h, R.ch, R.levels, R.pk["framerate"])
    frame_frac = round(b_alg * fps / world / (HBM_PEAK_GBS * 1e9), 5)
    rec = {"workload": "%s %dx%d, %d levels, 1 stream per GPU, %d frames per call%s" % (mode, R.w, R.h, R.levels, T, clip_tag),
           "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "us_per_frame": round(1e6 * dt / K, 2),
           "ms_per_step": round(1e3 * dt / K, 4), "frame_alg_bytes": b_alg, "frame_roofline_frac": frame_frac}
    if rank == 0 and R.keep >= R.n:
        from oracle import pyoracle as po
        nthreads = max(1, min(16, os.cpu_count() or 1))
        host = R.d_in[:, 0].cpu().numpy()
        # 4K: the oracle needs ~0.3 s per frame -- the checked frames are the first timed call's (the replay stops there)
        span = K if R.w * R.h <= 1920 * 1080 else min(K, T)
        check = sorted(set(int(round(x)) for x in np.linspace(base, base + span - 1, 8)))
        n_replay = check[-1] + 1
        keep, _, cdt = oracle_replay(po, np, host, R.pk, R.ring, n_replay, check, nthreads)
        dmax, fmin, ncmp = 0, 1.0, 0
        for i in check:
            ref, produced = keep[i]
            if produced:
                du = np.abs(ref.astype(np.int16) - R.d_out[i, 0].cpu().numpy().astype(np.int16))
                dmax, fmin, ncmp = max(dmax, int(du.max())), min(fmin, float((du == 0).mean())), ncmp + 1
        rec["verified"] = bool(ncmp >= 8 and dmax <= 1 and fmin >= 0.999)
        rec["verification"] = {"timed_frames_compared": ncmp, "u8_max_diff": dmax, "u8_identical_min": round(fmin, 6),
                               "bars": "u8 <= 1 LSB, >= 99.9 % identical", "oracle_frames_replayed": n_replay}
        rec["cpu_baseline"] = {"value": round(n_replay / cdt, 3), "unit": "frames/s", "cores": nthreads, "kind": "port", "host_cores": os.cpu_count(),
                               "sample": "%d frames of the same clip (the verification replay), CPU oracle = restatement of the reference, %d threads" % (n_replay, nthreads)}
        kernels, roofline = kernel_pass(lvm, torch, R, mode, 2 * T, RAMP_SECONDS)
        if roofline:
            roofline["frame_frac"] = frame_frac
            roofline["frame_frac_definition"] = "SURVEY.md 8d: B_alg x frames/s / peak (whole path, compulsory bytes only)"
        rec["roofline"] = roofline
        rec["kernels_us_per_launch"] = {k: v["avg_us"] for k, v in kernels.items()}
    host8 = R.d_in[:8, 0].cpu().numpy() if (extras and rank == 0) else None
    cp_keep = R.cp
    R.close()
    del R
    torch.cuda.empty_cache()
    if extras:
        rec["per_frame"] = per_frame_records(lvm, torch, np, cfg_idx, None, local_rank, ids, world, dist, red_dev)
        if rank == 0:
            ctx = lvm.Context(local_rank, 1)
            try:
                twin = lvm.load().lvm_optimal_buffer_size(int(pk["framerate"]))
                rec["e2e_host"] = e2e_host_record(lvm, np, ctx, C.byref(cp_keep), host8, Ke=(150 if host8.shape[1] <= 1080 else 40),
                                                  warm=(twin + 12 if mode == "color" else 20))
            finally:
                ctx.close()
    return rec


def sub_records(lvm, torch, np, args, cfg_idx, small, local_rank, rank, world, dist, red_dev):
    out = {}
    ids = lvm.sharding.stream_ids(rank, world, 1)
    # (1) per-frame schedule: K calls of lvm_process_device, the live one-frame-latency surface
    R = Runner(lvm, torch, np, cfg_idx, small, 1, 32, 1, local_rank, ids, 32)
    Kp = 300
    dt = timed_run(lvm, torch, R, Kp, 32, dist, red_dev)
    out["per_frame"] = {"schedule": "T = 1: one lvm_process_device call per frame, device-resident", "value": round(world * Kp / dt, 2),
                        "unit": "frames/s", "steps": Kp, "us_per_frame": round(1e6 * dt / Kp, 2),
                        "host_enqueue_us_per_frame": round(1e6 * lvm.sharding.timed_steps.host_seconds / Kp, 2)}
    try:        # kernel launches one frame costs on this surface (the profiler counts every LVM_LAUNCH)
        R.ctx.profile(True)
        R.run(20)
        torch.cuda.synchronize()
        prof = R.ctx.profile_collect()
        R.ctx.profile(False)
        out["per_frame"]["launches_per_frame"] = round(sum(v[1] for v in prof.values()) / 20.0, 2)
        out["per_frame"]["kernels_us"] = {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in prof.items()}
    except Exception as e:
        out["per_frame"]["launches_per_frame"] = None
        out["per_frame"]["profile_error"] = str(e)[:120]
    b_alg1 = lvm.load().lvm_algorithmic_bytes(R.pk["mode"], R.w, R.h, R.ch, R.levels, R.pk["framerate"])
    out["per_frame"]["frame_roofline_frac"] = round(b_alg1 * out["per_frame"]["value"] / world / (HBM_PEAK_GBS * 1e9), 5)
    # (1b) the same per-frame schedule with B streams in every launch (B in {1, 4, 16}, T = 1)
    pfs = {}
    for Bn in (4, 16):
        Rp = Runner(lvm, torch, np, cfg_idx, small, Bn, 16, 1, local_rank, ids, 16, time_shift=True)
        Kq = 120
        dtq = timed_run(lvm, torch, Rp, Kq, 16, dist, red_dev)
        v = world * Bn * Kq / dtq
        pfs["B%d" % Bn] = {"value": round(v, 2), "unit": "frames/s", "streams": Bn, "frames_per_call": 1, "us_per_frame": round(1e6 * dtq / (Kq * Bn), 3),
                           "frame_roofline_frac": round(b_alg1 * v / world / (HBM_PEAK_GBS * 1e9), 5)}
        if Bn == 4 and R.pk["mode"] == lvm.synth.MODE_LAPLACE:
            # the same calls with the library's depth-1 cross-frame pipeline (lvm_set_pipeline: the down-sweep of call t runs on a
            # second stream under the up-sweep of call t - 1; the output of call t is complete once call t + 1 -- or lvm_flush --
            # has been enqueued: a throughput schedule, NOT the one-call latency surface above)
            Rp.ctx.set_pipeline(1)
            dtp = timed_run(lvm, torch, Rp, Kq, 16, dist, red_dev)
            Rp.ctx.set_pipeline(0)
            vp = world * Bn * Kq / dtp
            pfs["B4_pipeline_depth1"] = {"value": round(vp, 2), "unit": "frames/s", "streams": Bn, "frames_per_call": 1, "us_per_frame": round(1e6 * dtp / (Kq * Bn), 3),
                                         "note": "output of call t complete when call t + 1 (or lvm_flush) is enqueued"}
        Rp.close()
        del Rp
        torch.cuda.empty_cache()
    out["per_frame_streams"] = pfs
    # (1c) the headline schedule once more at K = 400 with and without the clock ramp: what the driver's short timed region
    # (--steps 20 = one call) cannot show.  "cold" = the GPU ~2 ms out of idle when the timed region starts.
    Th = max(1, args.frames_per_call)
    hl = {}
    for name, rs in (("steady_state_k400", None), ("cold_k400", 0.0)):
        Rh = Runner(lvm, torch, np, cfg_idx, small, 1, ((32 + Th - 1) // Th) * Th, Th, local_rank, ids, 32)
        dth = timed_run(lvm, torch, Rh, 400, 64, dist, red_dev, ramp=rs)
        hl[name] = {"value": round(world * 400 / dth, 2), "unit": "frames/s", "steps": 400, "frames_per_call": Th,
                    "ramp_ms": (1e3 * RAMP_SECONDS if rs is None else 0.0), "us_per_frame": round(1e6 * dth / 400, 3)}
        Rh.close()
        del Rh
        torch.cuda.empty_cache()
    out["headline_run_length"] = hl
    # (2) host-to-host through lvm_process (the drop-in surface): pageable frames, then page-locked frames
    w, h, ch, fb = R.w, R.h, R.ch, R.frame_bytes
    host = R.d_in[:, 0].cpu().numpy()
    lib = lvm.load()
    twin = lib.lvm_optimal_buffer_size(int(R.pk["framerate"]))
    out["e2e_host"] = e2e_host_record(lvm, np, R.ctx, R.p_ref, host, kinds=("pageable", "pinned", "pageable_in_pinned_out", "pageable_staged_through_pinned"),
                                      warm=(twin + 12 if R.pk["mode"] == lvm.synth.MODE_COLOR else 20))
    # (2b) the export loop body: 32 page-locked host frames in, 32 side-by-side canvases out per lvm_export_frames call
    # (runChainOnce + Exporter::compose, Exporter.cpp:216-259; the temporal batch inside, the canvases composed on the device)
    out["export_host"] = export_host_record(lvm, np, local_rank, R.p_ref, host)
    R.close()
    del R
    torch.cuda.empty_cache()
    # (3) B streams per launch (time-shifted copies of the clip: independent temporal states)
    bs = {}
    for Bn in (4, 16):
        Tn = 32 if Bn == 4 else 16
        Rb = Runner(lvm, torch, np, cfg_idx, small, Bn, Tn, Tn, local_rank, ids, Tn, time_shift=True)
        Kb = 4 * Tn
        dtb = timed_run(lvm, torch, Rb, Kb, Tn, dist, red_dev)
        bs["B%d" % Bn] = {"value": round(world * Bn * Kb / dtb, 2), "unit": "frames/s", "streams": Bn, "frames_per_call": Tn,
                          "us_per_frame": round(1e6 * dtb / (Kb * Bn), 3)}
        Rb.close()
        del Rb
        torch.cuda.empty_cache()
    out["batched_streams"] = bs
    # (4) BASELINE configs[4]
    out["cfg4_riesz_4k"] = config_record(lvm, torch, np, 4, local_rank, rank, world, dist, red_dev, K=64, W=16, T=16, ring=16)
    # (5) BASELINE configs[2] and configs[3] (the headline run only: `--mode riesz|color` runs them as the main line)
    if cfg_idx == 1 and not small:
        out["cfg2_riesz_1080p"] = config_record(lvm, torch, np, 2, local_rank, rank, world, dist, red_dev)
        out["cfg3_color_1080p"] = config_record(lvm, torch, np, 3, local_rank, rank, world, dist, red_dev)
        # (6) content sensitivity: the headline clip carries +-12 levels of per-channel noise -- neighbouring pixels land in different
        # cells of the forward Lab table, the worst case for its gathers; the same stream with a camera's +-2 levels
        global CLIP_NOISE
        if CLIP_NOISE is None:
            CLIP_NOISE = 2.0
            try:
                rec = config_record(lvm, torch, np, 1, local_rank, rank, world, dist, red_dev, K=128, extras=False,
                                    clip_tag=", clip noise +-2 levels instead of the headline's +-12")
            finally:
                CLIP_NOISE = None
            out["camera_noise_clip"] = rec
    return out


if __name__ == "__main__":
    main()
