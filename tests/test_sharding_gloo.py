"""N > 1 path on CPU: world_size 2 over gloo.  The hot path shards by stream with no data-path
collective; what the ranks share is the barrier + MAX-reduce timing protocol and the disjoint
stream/seed assignment that bench.py uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import importlib
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    lvm = importlib.import_module("live-video-magnification_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = lvm.sharding.stream_ids(rank, world, 3)
    seeds = [lvm.sharding.stream_seed(i) for i in ids]
    # each rank "processes" its own frames; rank 1 is deliberately slower
    frames = {"n": 0}

    def step(i):
        time.sleep(0.002 * (1 + 2 * rank))
        frames["n"] += 3
    dt = lvm.sharding.timed_steps(step, 10, dist)
    fps = lvm.sharding.aggregate_fps(world, 3, 10, dt)
    # the clips of different ranks differ, the clips of one rank are reproducible
    f0 = lvm.synth.Clip(32, 24, seed=seeds[0]).frame(1)
    gathered = [None] * world
    dist.all_gather_object(gathered, (ids, seeds, int(f0.sum()), frames["n"]))
    q.put((rank, dt, fps, gathered))
    dist.destroy_process_group()


def test_two_rank_stream_sharding_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, fps0, g0), (r1, dt1, fps1, g1) = res
    assert dt0 == dt1                          # MAX over ranks is what every rank reports
    assert dt0 >= 10 * 0.006 * 0.9             # the slow rank (6 ms per step) sets the time
    assert fps0 == fps1 == pytest.approx(2 * 3 * 10 / dt0)
    ids = g0[0][0] + g0[1][0]
    assert sorted(ids) == list(range(6)) and len(set(ids)) == 6      # disjoint, complete
    assert len({g0[0][2], g0[1][2]}) == 2                            # different streams => different frames
    assert g0[0][3] == g0[1][3] == 30                                # fixed per-rank work (weak scaling)


def test_stream_ids_cover_all_streams():
    import importlib
    lvm = importlib.import_module("live-video-magnification_amd")
    for world in (1, 2, 4, 8):
        allids = sum((lvm.sharding.stream_ids(r, world, 2) for r in range(world)), [])
        assert allids == list(range(2 * world))
