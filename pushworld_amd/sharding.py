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


def c4_global_puzzle_ids(total_envs: int, n_level0: int, n_higher: int, seed: int = 100):
    """Puzzle of every environment of config C4 over the WHOLE job, independent of the number of
    ranks: even environments draw uniformly from the ``n_level0`` Level-0 train puzzles (pool indices
    0 .. n_level0 - 1), odd ones from the ``n_higher`` Level-1..4 puzzles (indices n_level0 ..) -- a 50 / 50 mix.
    Every rank computes the same array and keeps its ``shard_puzzle_ids`` slice (sorted inside the shard so that
    environments of one puzzle sit together)."""
    import numpy as np

    if total_envs < 0 or n_level0 < 1 or n_higher < 1:
        raise ValueError("c4_global_puzzle_ids: empty puzzle pool")
    rng = np.random.Generator(np.random.PCG64(seed))
    lo = rng.integers(0, n_level0, size=total_envs, dtype=np.int64)
    hi = rng.integers(0, n_higher, size=total_envs, dtype=np.int64) + n_level0
    return np.where(np.arange(total_envs) % 2 == 0, lo, hi)


def _group():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


def reduce_max(values, device=None):
    """Element-wise MAX over all ranks of a list of floats (identity without a process group)."""
    dist = _group()
    if dist is None:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def gather_floats(value: float, device=None):
    """One float of every rank, in rank order (``[value]`` without a process group)."""
    dist = _group()
    if dist is None:
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def gather_vectors(values, device=None):
    """A fixed-length list of floats of every rank, in rank order (``[values]`` without a process group)."""
    dist = _group()
    if dist is None:
        return [[float(v) for v in values]]
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(v) for v in t.tolist()] for t in out]


def device_numa_cpus(device_index: int):
    """(NUMA node, CPU set) local to a HIP device, from its PCI address in sysfs; (None, None) when unknown."""
    import os

    try:
        props = torch.cuda.get_device_properties(device_index)
        addr = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        base = os.path.join("/sys/bus/pci/devices", addr)
        with open(os.path.join(base, "numa_node")) as f:
            node = int(f.read().strip())
        with open(os.path.join(base, "local_cpulist")) as f:
            text = f.read().strip()
        cpus = set()
        for part in text.split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return (node if node >= 0 else None), (cpus or None)
    except Exception:  # noqa: BLE001 -- no sysfs entry, old torch without the PCI fields, ...
        return None, None


def pin_to_device_numa(device_index: int):
    """Pins the calling thread (the one that launches the kernels: the launch rate is host bound at ~8 us per
    launch) to the CPUs of the device's NUMA node.  Returns (node or None, previous affinity mask or None)."""
    import os

    node, cpus = device_numa_cpus(device_index)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return None, None
    try:
        before = os.sched_getaffinity(0)
        allowed = cpus & before
        if not allowed:
            return None, None
        os.sched_setaffinity(0, allowed)
        return node, before
    except OSError:
        return None, None


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
