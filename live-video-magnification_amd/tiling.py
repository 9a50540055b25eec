"""Spatial tiling of ONE Riesz stream over several ranks -- a correctness demonstrator (north_star's "optional single-4K
spatial-tile halo over RCCL"; include/lvm_hip.h lvm_tile_riesz_*).  NOT the production answer to several GPUs: a frame costs a gather and a
scatter of small planes -- latencies -- against ~230 us of work for a whole 4K frame on one MI355X; independent streams, one per GPU, scale
without any exchange (sharding.py).  What it shows is that the path CAN be cut spatially with bit-identical results.

The frame is cut into `world` horizontal stripes whose boundaries are multiples of 2^F (F = fine levels, default 2).  Rank r owns rows
[own0, own1) and computes on [own0 - halo, own1 + halo) clipped to the frame -- the reference's per-frame stages, stencil by stencil,
reach at most 40 level-0 rows beyond a row they produce when the levels >= F come from elsewhere (build 9x9: 4 rows per level; Riesz pair 2;
the two 13-tap blurs 6 + 6; collapse 9x9: 4 -- accumulated over levels 0 and 1), so with halo = 64 the owned rows never see an artificial
stripe edge.  Per frame:

    every rank   stage 1   Lab, pyramid levels 0 .. F-1, phase, temporal filters, amplify on its extended stripe   (lvm_tile_riesz_stage1)
                 --------  owned rows of octave F -> rank 0                                                         [exchange 1: gather]
    rank 0       coarse    levels F .. L-1 on the assembled octave F, collapsed to res_F                            (lvm_tile_riesz_planes)
                 --------  rows of res_F covering each rank's extended stripe -> that rank                          [exchange 2: scatter]
    every rank   stage 2   collapse of levels F-1 .. 0 from those rows, Lab2BGR, u8; keep the owned rows            (lvm_tile_riesz_stage2)

Exchanges go through torch.distributed: backend "nccl" (= RCCL over xGMI, device tensors, one rank per GPU) or "gloo" (host tensors; tests:
two ranks sharing one GPU, or the CPU emulation build).  Memory is a small array layer so that the same code drives numpy buffers (the
emulation build's "device" memory is host memory) and torch tensors on a GPU.
"""
import ctypes as C

import numpy as np


def stripe_plan(h, world, fine_levels=2, halo=64):
    """[(own0, own1, ext0, ext1)] per rank: boundaries at multiples of 2^F, extended by the halo, clipped to the frame"""
    unit = 1 << fine_levels
    if halo % unit or halo < 40:
        raise ValueError("halo: a multiple of 2^F, at least 40 rows (the reach of the fine levels' stencils)")
    bounds = [0]
    for r in range(1, world):
        bounds.append(min(h, max(bounds[-1] + unit, (h * r // world) // unit * unit)))
    bounds.append(h)
    plan = []
    for r in range(world):
        own0, own1 = bounds[r], bounds[r + 1]
        if own1 <= own0:
            raise ValueError("more ranks than stripes of %d rows" % unit)
        plan.append((own0, own1, max(0, own0 - halo), min(h, own1 + halo)))
    return plan


def level_rows(n, levels):
    """rows (or columns) of pyramid level `levels` of a plane with n rows: the reference halves with (n + 1) / 2"""
    for _ in range(levels):
        n = (n + 1) // 2
    return n


class _Numpy:
    """buffers of the emulation build (device memory = host memory)"""
    def empty(self, shape, dtype):
        return np.empty(shape, dtype)

    def ptr(self, a):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data

    def from_host(self, a):
        return np.ascontiguousarray(a)

    def to_host(self, a):
        return np.array(a, copy=True)

    def copy_rows(self, dst, d0, src, s0, n):
        dst[d0:d0 + n] = src[s0:s0 + n]

    def sync(self, ctx):
        ctx.synchronize()


class _Torch:
    """buffers on a GPU"""
    def __init__(self, torch, device):
        self.torch, self.device = torch, torch.device("cuda", device)
        self._dt = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32}

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=self._dt[np.dtype(dtype)], device=self.device)

    def ptr(self, a):
        assert a.is_contiguous()
        return a.data_ptr()

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def to_host(self, a):
        return a.cpu().numpy()

    def copy_rows(self, dst, d0, src, s0, n):
        dst[d0:d0 + n].copy_(src[s0:s0 + n])

    def sync(self, ctx):
        ctx.synchronize()
        self.torch.cuda.synchronize(self.device)


class StripeWorker:
    """One rank's stripe context (+ the coarse context on rank 0)."""

    def __init__(self, lvm, rank, world, w, h, pk, fine_levels=2, halo=64, lib=None, device=0, mem=None, exact=False):
        self.lvm, self.rank, self.world, self.w, self.h, self.F = lvm, rank, world, w, h, fine_levels
        self.plan = stripe_plan(h, world, fine_levels, halo)
        self.own0, self.own1, self.ext0, self.ext1 = self.plan[rank]
        self.mem = mem or _Numpy()
        L = int(pk["levels"])
        if L < fine_levels + 2:
            raise ValueError("tiling needs at least two levels above the fine ones")
        self.lib = lib or lvm.load()
        mk = lambda lv: lvm.LvmParams(pk["mode"], lv, pk["amplification"], pk["coWavelength"], pk["coLow"], pk["coHigh"], pk["chromAttenuation"], pk["framerate"], 0)   # noqa: E731
        self.p_fine, self.p_coarse = mk(fine_levels + 1), mk(L - fine_levels)
        self.ctx = lvm.Context(device, 1, self.lib)
        self.ctx.exact_lab(exact)
        self.eh = self.ext1 - self.ext0
        self.rw, self.rh = level_rows(w, fine_levels), level_rows(self.eh, fine_levels)
        self.residual = self.mem.empty((self.rh, self.rw), np.float32)           # octave F of the extended stripe
        self.res_in = self.mem.empty((self.rh, self.rw), np.float32)             # rows of res_F for the extended stripe
        self.out = self.mem.empty((self.eh, w, 3), np.uint8)
        self.unit = 1 << fine_levels
        # rows of octave F: owned / extended, in full-plane and in local coordinates
        self.f_ext0 = self.ext0 // self.unit
        self.f_own0 = self.own0 // self.unit
        self.f_own1 = level_rows(h, fine_levels) if self.own1 == h else self.own1 // self.unit
        self.coarse = None
        if rank == 0:
            self.hF = level_rows(h, fine_levels)
            self.coarse = lvm.Context(device, 1, self.lib)
            self.coarse.exact_lab(exact)
            self.octF = self.mem.empty((self.hF, self.rw), np.float32)
            self.resF = self.mem.empty((self.hF, self.rw), np.float32)

    def close(self):
        self.ctx.close()
        if self.coarse:
            self.coarse.close()

    # ---- the three calls -----------------------------------------------------------------------------------------------------------------
    def stage1(self, d_frame_ext):
        """d_frame_ext: the rows [ext0, ext1) of the frame in device memory, (eh, w, 3) u8.  -> produced"""
        produced, rw, rh = C.c_int(0), C.c_int(0), C.c_int(0)
        self.ctx._check(self.lib.lvm_tile_riesz_stage1(self.ctx.h, C.byref(self.p_fine), self.mem.ptr(d_frame_ext), self.w, self.eh, self.w * 3, C.byref(produced),
                                                       self.mem.ptr(self.residual), C.byref(rw), C.byref(rh), None))
        assert (rw.value, rh.value) == (self.rw, self.rh), ((rw.value, rh.value), (self.rw, self.rh))
        self.mem.sync(self.ctx)
        return bool(produced.value)

    def owned_residual_rows(self):
        """(first full-plane row of octave F, the owned rows of this stripe's residual octave)"""
        a = self.f_own0 - self.f_ext0
        return self.f_own0, self.residual[a:a + (self.f_own1 - self.f_own0)]

    def coarse_planes(self):
        """rank 0: levels F .. L-1 on the assembled octave F -> res_F.  -> produced"""
        produced = C.c_int(0)
        self.coarse._check(self.lib.lvm_tile_riesz_planes(self.coarse.h, C.byref(self.p_coarse), self.mem.ptr(self.octF), self.rw, self.hF, self.mem.ptr(self.resF),
                                                          C.byref(produced), None))
        self.mem.sync(self.coarse)
        return bool(produced.value)

    def stage2(self, d_frame_ext):
        """collapse from self.res_in, Lab2BGR, u8 -> the owned rows (device buffer view)"""
        self.ctx._check(self.lib.lvm_tile_riesz_stage2(self.ctx.h, C.byref(self.p_fine), self.mem.ptr(d_frame_ext), self.w, self.eh, self.w * 3, self.mem.ptr(self.res_in),
                                                       self.mem.ptr(self.out), self.w * 3, None))
        self.mem.sync(self.ctx)
        return self.out[self.own0 - self.ext0:self.own1 - self.ext0]


def run_local(lvm, frames, pk, world=2, fine_levels=2, halo=64, lib=None, mem=None, exact=False):
    """All ranks in ONE process (no torch.distributed): the exchanges are row copies.  frames: list of (h, w, 3) u8 host arrays.
    -> list of (frame u8 host array, produced)."""
    h, w = frames[0].shape[:2]
    mem = mem or _Numpy()
    workers = [StripeWorker(lvm, r, world, w, h, pk, fine_levels, halo, lib, 0, mem, exact) for r in range(world)]
    root = workers[0]
    out = []
    try:
        for f in frames:
            ext = [mem.from_host(f[wk.ext0:wk.ext1]) for wk in workers]
            flags = [wk.stage1(e) for wk, e in zip(workers, ext)]
            for wk in workers:                                           # exchange 1: gather the owned rows of octave F
                row0, rows = wk.owned_residual_rows()
                mem.copy_rows(root.octF, row0, rows, 0, rows.shape[0])
            pc = root.coarse_planes()
            assert all(x == flags[0] for x in flags) and pc == flags[0]
            if not flags[0]:
                out.append((f.copy(), False))                            # MagnificationProcessor.cpp:61: the input frame
                continue
            res = np.empty_like(f)
            for wk, e in zip(workers, ext):                              # exchange 2: scatter rows of res_F
                mem.copy_rows(wk.res_in, 0, root.resF, wk.f_ext0, wk.rh)
                res[wk.own0:wk.own1] = mem.to_host(wk.stage2(e))
            out.append((res, True))
    finally:
        for wk in workers:
            wk.close()
    return out


def run_rank(lvm, dist, rank, world, frames, pk, fine_levels=2, halo=64, lib=None, device=0, use_gpu=False, exact=False):
    """One rank of `world` under torch.distributed (already initialised): exchanges by send / recv -- device tensors over "nccl" (RCCL),
    host tensors over "gloo".  Every rank is given the whole frames here (a test harness; a deployment uploads only the extended stripe).
    -> on rank 0: list of (frame, produced); elsewhere: None."""
    import torch
    nccl = dist.get_backend() == "nccl"
    mem = _Torch(torch, device) if use_gpu else _Numpy()
    h, w = frames[0].shape[:2]
    wk = StripeWorker(lvm, rank, world, w, h, pk, fine_levels, halo, lib, device, mem, exact)
    plan = wk.plan
    unit = 1 << fine_levels

    def as_msg(a):        # what dist can move: a device tensor over nccl, a host tensor over gloo
        if use_gpu:
            return a.contiguous() if nccl else a.cpu().contiguous()
        return torch.from_numpy(np.ascontiguousarray(a))

    def from_msg(t, like):
        if use_gpu:
            return t if nccl else t.to(like.device)
        return t.numpy()

    def msg_empty(rows):
        return torch.empty((rows, wk.rw), dtype=torch.float32, device=(mem.device if (use_gpu and nccl) else "cpu"))
    out = []
    try:
        for f in frames:
            ext = mem.from_host(f[wk.ext0:wk.ext1])
            produced = wk.stage1(ext)
            # exchange 1: owned rows of octave F -> rank 0
            row0, rows = wk.owned_residual_rows()
            if rank == 0:
                mem.copy_rows(wk.octF, row0, rows, 0, rows.shape[0])
                for r in range(1, world):
                    o0, o1 = plan[r][0], plan[r][1]
                    a = o0 // unit
                    b = wk.hF if o1 == h else o1 // unit
                    t = msg_empty(b - a)
                    dist.recv(t, src=r)
                    mem.copy_rows(wk.octF, a, from_msg(t, wk.octF), 0, b - a)
                wk.coarse_planes()
                # exchange 2: rows of res_F for every rank's extended stripe
                mem.copy_rows(wk.res_in, 0, wk.resF, wk.f_ext0, wk.rh)
                for r in range(1, world):
                    e0, e1 = plan[r][2], plan[r][3]
                    n = level_rows(e1 - e0, fine_levels)
                    dist.send(as_msg(wk.resF[e0 // unit:e0 // unit + n]), dst=r)
            else:
                dist.send(as_msg(rows), dst=0)
                t = msg_empty(wk.rh)
                dist.recv(t, src=0)
                mem.copy_rows(wk.res_in, 0, from_msg(t, wk.res_in), 0, wk.rh)
            mine = mem.to_host(wk.stage2(ext)) if produced else f[wk.own0:wk.own1].copy()
            # the result: stripes to rank 0 (test harness; u8 rows as host tensors)
            gathered = [None] * world if rank == 0 else None
            dist.gather_object((wk.own0, wk.own1, mine, produced), gathered, dst=0)
            if rank == 0:
                res = np.empty_like(f)
                for o0, o1, rows_u8, _ in gathered:
                    res[o0:o1] = rows_u8
                assert len({g[3] for g in gathered}) == 1
                out.append((res, produced))
    finally:
        wk.close()
    return out if rank == 0 else None
