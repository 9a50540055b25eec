"""Stream sharding across GPUs: the hot path partitions by stream -- every stream owns
its temporal state and never exchanges data with another -- so N ranks run disjoint stream sets and
the only cross-rank traffic is the timing barrier and one MAX-reduce of the elapsed time.  No
data-path collective exists."""
import time


def stream_ids(rank, world, streams_per_rank):
    """Global ids of the streams rank `rank` owns (weak scaling: per-rank work is fixed)."""
    assert 0 <= rank < world and streams_per_rank >= 1
    return list(range(rank * streams_per_rank, (rank + 1) * streams_per_rank))


def stream_seed(stream_id, base=1234):
    """Seed of the synthetic clip of a stream (seeds 1234..)."""
    return base + stream_id


def barrier(dist, sync=None):
    if sync:
        sync()
    if dist is not None:
        dist.barrier()
    if sync:
        sync()


def timed_steps(step, steps, dist=None, sync=None, device=None, finish=None):
    """Runs `steps` calls of step(i) bracketed by barrier + device sync on both sides and returns
    the MAX elapsed seconds over all ranks (the job is as slow as its slowest rank)."""
    import torch
    barrier(dist, sync)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    if finish:
        finish()          # e.g. drain a software pipeline: its work belongs to the timed region
    timed_steps.host_seconds = time.perf_counter() - t0   # host-side enqueue time (before the device drains)
    if sync:
        sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    barrier(dist, sync)
    return dt


def aggregate_fps(world, streams_per_rank, steps, max_seconds):
    """Whole-job throughput: every rank processed streams_per_rank * steps frames."""
    return world * streams_per_rank * steps / max_seconds
