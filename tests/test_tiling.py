"""Spatial tiling of ONE Riesz stream (north_star's optional single-4K spatial-tile halo): tiling.py + lvm_tile_riesz_*.
A correctness demonstrator: stripes with a 64-row halo, the coarse levels gathered on rank 0, results BIT-IDENTICAL to the unsplit context.
CPU: the emulation build, all ranks in one process and two ranks over gloo.  GPU: the gfx950 library, two and three stripes in one process
and two ranks sharing the one GPU of the box over gloo (RCCL refuses two ranks on one device; with a GPU per rank the same code moves
device tensors over "nccl")."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from helpers import c_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unsplit(lvm, lib, frames, pk, exact):
    ctx = lvm.Context(0, 1, lib)
    ctx.exact_lab(exact)
    cp = c_params(lvm, pk)
    out = []
    try:
        for f in frames:
            o, pr = ctx.process(f, cp)
            out.append((o.copy(), pr))
    finally:
        ctx.close()
    return out


def _clip(lvm, w, h, levels, n):
    ck, pk = lvm.synth.config(2, (w, h, levels))
    clip = lvm.synth.Clip(**ck)
    return [clip.frame(t) for t in range(n)], pk


def _same(a, b):
    assert len(a) == len(b)
    for t, ((fa, pa), (fb, pb)) in enumerate(zip(a, b)):
        assert pa == pb, t
        bad = np.argwhere(fa != fb)
        assert len(bad) == 0, (t, len(bad), bad[:4].tolist())


def test_stripe_plan():
    import importlib
    lvm = importlib.import_module("live-video-magnification_amd")
    plan = lvm.tiling.stripe_plan(2160, 2)
    assert plan == [(0, 1080, 0, 1144), (1080, 2160, 1016, 2160)]
    plan = lvm.tiling.stripe_plan(360, 3)
    assert [p[:2] for p in plan] == [(0, 120), (120, 240), (240, 360)] and plan[1][2:] == (56, 304)
    with pytest.raises(ValueError):
        lvm.tiling.stripe_plan(360, 2, halo=32)            # under the reach of the stencils


@pytest.mark.parametrize("w,h,levels,world", [(192, 256, 5, 2), (160, 392, 5, 3), (128, 200, 4, 2)])
def test_tiled_riesz_equals_the_unsplit_context_emu(lvm, emu, w, h, levels, world):
    """stripes + gathered coarse levels on the emulation build, all ranks in one process: every byte of every frame equal to the unsplit
    context's (exact flavour: itself bit-identical to the oracle, tests/test_emu_parity.py).  392 = 3 stripes with a clipped last one;
    200 rows: the last stripe's height is not a multiple of 4."""
    frames, pk = _clip(lvm, w, h, levels, 5)
    want = _unsplit(lvm, emu, frames, pk, True)
    got = lvm.tiling.run_local(lvm, frames, pk, world=world, lib=emu, exact=True)
    _same(got, want)
    assert [p for _, p in got] == [False, True, True, True, True]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, q, lib_path, use_gpu, w, h, levels, n):
    import ctypes
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    lvm = importlib.import_module("live-video-magnification_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = lvm.bind(ctypes.CDLL(lib_path)) if lib_path else lvm.load()
    frames, pk = _clip(lvm, w, h, levels, n)
    out = lvm.tiling.run_rank(lvm, dist, rank, world, frames, pk, lib=lib, use_gpu=use_gpu, exact=not use_gpu)
    if rank == 0:
        want = _unsplit(lvm, lib, frames, pk, not use_gpu)
        diff = [int((a != b).sum()) for (a, _), (b, _) in zip(out, want)]
        q.put((diff, [p for _, p in out], [p for _, p in want]))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(lib_path, use_gpu, w, h, levels, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q, lib_path, use_gpu, w, h, levels, n)) for r in range(2)]
    for p in procs:
        p.start()
    diff, pa, pb = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert pa == pb and diff == [0] * n, (diff, pa, pb)


def test_tiled_riesz_two_ranks_over_gloo_emu(emu):
    """world_size 2 over gloo, the emulation build on every rank: the gather of octave 2 and the scatter of res_2 through send / recv"""
    _two_ranks(os.path.join(ROOT, "tests", "emu", "_build", "liblvm_emu.so"), False, 192, 256, 5, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,levels,world,n", [(3840, 2160, 8, 2, 4), (1920, 1080, 6, 3, 4)])
def test_tiled_riesz_equals_the_unsplit_context_gpu(lvm, hip, w, h, levels, world, n):
    """BASELINE configs[4]'s frame (3840 x 2160, 8 levels) in two stripes, 1080p in three: the gfx950 library's default flavour, u8 frames
    bit-equal to the unsplit context's"""
    import torch
    frames, pk = _clip(lvm, w, h, levels, n)
    want = _unsplit(lvm, hip, frames, pk, False)
    got = lvm.tiling.run_local(lvm, frames, pk, world=world, lib=hip, mem=lvm.tiling._Torch(torch, 0))
    _same(got, want)


@pytest.mark.gpu
def test_tiled_riesz_two_ranks_share_the_gpu_over_gloo():
    _two_ranks(None, True, 1920, 1080, 6, 3)
