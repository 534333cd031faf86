"""Multi-GPU scheme of the engine: ONE process per GPU, each owning a full model replica and a disjoint
round-robin share of the key frames; no data-path collective (every key frame + its sources is an
independent forward).  The process group is only used to line the ranks up for timing.
(The reference's equivalent is nn.DataParallel, rmvd/models/helpers.py:161-169, which re-broadcasts the
weights on every call.)"""
import time


def frames_for_rank(num_frames, rank, world_size):
    """Round-robin shard: rank r processes frames r, r + W, r + 2W, ..."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside 0..{world_size - 1}")
    return list(range(rank, num_frames, world_size))


def timed_region(fn, sync, dist=None, device=None):
    """Runs fn() bracketed by barrier + device sync on both sides; returns the MAX wall time over ranks."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt
