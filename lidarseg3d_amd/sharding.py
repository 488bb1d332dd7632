"""Frame sharding across GPUs.

Every stage of the path is per-frame: frames are independent units, so N GPUs run N replicas of the
forward, one process per GPU, each on its own share of the frames — no collective on the data path.  The only
communication is bookkeeping: a barrier around timed regions and a MAX-reduction of per-rank elapsed times (bench.py),
and, for evaluation, gathering per-frame predictions (the reference pickles and all_gathers them,
det3d/torchie/trainer/utils.py:114-154).  `backend="nccl"` is RCCL on ROCm; the same code runs on gloo for CPU tests."""
import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_frames(n_frames, rank, world):
    """contiguous, balanced share of frame indices for `rank` (first n_frames % world ranks get one extra)"""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (no-op without an initialised process group)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_frame_results(local, device="cpu"):
    """all ranks -> list (ordered by rank) of per-rank python objects; frames come back in global frame order when every
    rank passes the results of shard_frames(...) in order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    return out
