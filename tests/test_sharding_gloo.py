"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding and the counter
reduction used by bench.py (the GPU box runs the same code over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from pushworld_amd.sharding import reduce_counters, shard_bounds, shard_puzzle_ids

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = (np.arange(total) * 68) // total
    mine = shard_puzzle_ids(ids, rank, world)
    lo, hi = shard_bounds(total, rank, world)
    counters = {"env_steps": (hi - lo) * 10, "episodes": rank + 1, "first_id": int(mine[0]) if len(mine) else 0}
    summed, elapsed = reduce_counters(counters, 0.5 + rank)
    dist.barrier()
    np.save(os.path.join(out_dir, f"r{rank}.npy"),
            np.array([summed["env_steps"], summed["episodes"], elapsed, lo, hi, len(mine)], dtype=np.float64))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [65536 * 2, 1001])
def test_two_rank_gloo_sharding(tmp_path, total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    for r, row in enumerate(rows):
        assert row[0] == total * 10 and row[1] == 3 and row[2] == 1.5  # SUM, SUM, MAX over ranks
    assert rows[0][3] == 0 and rows[0][4] == rows[1][3] and rows[1][4] == total  # contiguous cover
    assert abs(rows[0][5] - rows[1][5]) <= 1


def test_shard_bounds_cover_everything():
    from pushworld_amd.sharding import reduce_counters, shard_bounds

    for total in (0, 1, 7, 65536, 524288):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert reduce_counters({"a": 3}, 2.0) == ({"a": 3}, 2.0)  # no process group: identity
