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

    from pushworld_amd.sharding import (c4_global_puzzle_ids, gather_floats, gather_vectors, reduce_counters, reduce_max,
                                        shard_bounds, shard_puzzle_ids)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = (np.arange(total) * 68) // total
    mine = shard_puzzle_ids(ids, rank, world)
    lo, hi = shard_bounds(total, rank, world)
    # the vector bench.py reduces: the four device-side counters of pw_counters (+ the rank count)
    counters = {"env_steps": (hi - lo) * 10, "episodes_ended": rank + 1, "episodes_solved": rank, "bad_actions": 0, "ranks": 1}
    summed, elapsed = reduce_counters(counters, 0.5 + rank)
    # bench.py's per-window MAX over ranks and per-rank report
    wmax = reduce_max([1.0 + rank, 5.0 - rank])
    per_rank = gather_floats(10.0 * (rank + 1))
    rows = gather_vectors([rank + 0.5, 7.0, -1.0 - rank])  # bench.py's per-rank report rows
    assert rows == [[0.5, 7.0, -1.0], [1.5, 7.0, -2.0]], rows
    # config C4: every rank derives the same global assignment and keeps its own slice
    glob = c4_global_puzzle_ids(total, 14000, 223, 100)
    c4_mine = np.sort(shard_puzzle_ids(glob, rank, world))
    dist.barrier()
    np.save(os.path.join(out_dir, f"r{rank}.npy"),
            np.array([summed["env_steps"], summed["episodes_ended"] + 10 * summed["episodes_solved"] + 100 * summed["ranks"]
                      + 1000 * summed["bad_actions"], elapsed, lo, hi, len(mine), wmax[0], wmax[1],
                      per_rank[0], per_rank[1], float(glob.sum()), float(c4_mine.sum()), float(len(c4_mine)),
                      float((c4_mine < 14000).sum())], dtype=np.float64))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [65536 * 2, 1001])
def test_two_rank_gloo_sharding(tmp_path, total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    for r, row in enumerate(rows):
        assert row[0] == total * 10 and row[1] == 3 + 10 * 1 + 100 * 2 and row[2] == 1.5  # SUM (5-vector), MAX over ranks
    assert rows[0][3] == 0 and rows[0][4] == rows[1][3] and rows[1][4] == total  # contiguous cover
    assert abs(rows[0][5] - rows[1][5]) <= 1
    for row in rows:
        assert row[6] == 2.0 and row[7] == 5.0                 # element-wise MAX over ranks
        assert row[8] == 10.0 and row[9] == 20.0               # gathered in rank order
    assert rows[0][10] == rows[1][10]                           # same global C4 assignment on both ranks
    assert rows[0][11] + rows[1][11] == rows[0][10] and rows[0][12] + rows[1][12] == total  # the shards partition it
    for row in rows:
        assert abs(row[13] / row[12] - 0.5) < 0.01              # 50 % Level 0 inside every shard


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


def test_c4_assignment_is_rank_independent_and_balanced():
    from pushworld_amd.sharding import c4_global_puzzle_ids, gather_floats, reduce_max, shard_puzzle_ids

    g = c4_global_puzzle_ids(524288, 14000, 223, 100)
    assert g.shape == (524288,) and (g[0::2] < 14000).all() and (g[1::2] >= 14000).all() and g.max() < 14223
    assert (g == c4_global_puzzle_ids(524288, 14000, 223, 100)).all()
    assert len(np.unique(g[0::2])) == 14000 and len(np.unique(g[1::2])) == 223   # every puzzle of the mix is used
    parts = [shard_puzzle_ids(g, r, 8) for r in range(8)]
    assert all(len(p) == 65536 for p in parts) and (np.concatenate(parts) == g).all()
    assert reduce_max([1.0, 2.0]) == [1.0, 2.0] and gather_floats(3.0) == [3.0]
    from pushworld_amd.sharding import gather_vectors, pin_to_device_numa
    assert gather_vectors([1.0, 2.5]) == [[1.0, 2.5]]       # no process group: this rank only
    assert pin_to_device_numa(0) == (None, None)             # no HIP device here: nothing is pinned
    with pytest.raises(ValueError):
        c4_global_puzzle_ids(10, 0, 223)


def test_bench_refuses_more_ranks_than_devices():
    """No GPU here: ``bench.py --gpus 8`` must exit non-zero and print no result line (it used to run one rank and
    report n_gpus 1)."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2"],
                          capture_output=True, text=True, env=env, timeout=300)
    assert proc.returncode != 0
    assert '"metric"' not in proc.stdout
