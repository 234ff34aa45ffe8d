"""Multi-GPU plumbing for the bench / data-parallel drivers.

The hot path shards by independent clouds: every rank samples,
groups and interpolates its own clouds and there is NO data-path collective.  The only
communication is control-plane: a barrier around the timed region and a MAX over ranks of
the elapsed time.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return world, rank, local


def cloud_range(rank, clouds_per_rank, total=None):
    """Global indices [start, stop) of the clouds rank `rank` owns: contiguous blocks of `clouds_per_rank` in rank order
    (what exchange_anchor_rows' lowest-rank tie rule assumes), clipped to `total` clouds when given -- the last ranks of an
    uneven split own fewer (possibly zero) clouds."""
    lo, hi = rank * clouds_per_rank, (rank + 1) * clouds_per_rank
    if total is not None:
        lo, hi = min(lo, total), min(hi, total)
    return lo, hi


def gather_over_ranks(value, device="cpu"):
    """[value of rank 0, ..., value of rank W-1] on every rank (a float per rank; control plane only)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
