"""Multi-GPU layout of the batched engine: one process per GPU, environments sharded by rank.

Environments are independent (the reference has no cross-environment state at all), so the
data path needs NO collective: every rank owns a contiguous shard of the environment index
space with its own engine, stream and action stream.  ``torch.distributed`` (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only to agree on the timing
window and to sum throughput counters -- a few bytes per measurement window.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def shard_bounds(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) shard of ``total_envs`` for ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_puzzle_ids(puzzle_ids, rank: int, world: int):
    """The rank's slice of a global puzzle-id assignment (kept grouped by puzzle)."""
    lo, hi = shard_bounds(len(puzzle_ids), rank, world)
    return puzzle_ids[lo:hi]


def reduce_counters(counters: Dict[str, int], elapsed_s: float, device=None) -> Tuple[Dict[str, int], float]:
    """SUM of integer counters and MAX of the elapsed time over all ranks (identity when
    torch.distributed is not initialised)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return dict(counters), float(elapsed_s)
    keys = sorted(counters)
    vals = torch.tensor([int(counters[k]) for k in keys], dtype=torch.int64, device=device)
    el = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(vals, op=dist.ReduceOp.SUM)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return {k: int(v) for k, v in zip(keys, vals.tolist())}, float(el.item())
