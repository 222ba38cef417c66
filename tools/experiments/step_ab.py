#!/usr/bin/env python3
"""A/B of the lane-group step kernel (state only), one process per library build:

  * PW_OPT_STEP_LDS_TABLES: the puzzle's wall / agent-wall / shape row bitboards staged in LDS per lane group
    (what BASELINE.json's north_star sketches) against the default (rows read from global memory through L1);
  * builds with __launch_bounds__(256, W) (PUSHWORLD_AMD_LIB=tools/experiments/bin/libpw_wavesW.so).

Workloads: C2 (4 096 copies of one Level-0 puzzle), C3 (65 536 Level-1 environments grouped by puzzle), the C4
shard (65 536 environments over all 14 223 puzzles, N_pad 32); one step per launch and 64-step rollouts.
Output: microseconds per launch (median of N), env-steps/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pushworld_amd import _capi  # noqa: E402
from pushworld_amd import benchmark_data as bd  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.sharding import c4_global_puzzle_ids, shard_puzzle_ids  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402


def timed(fn, n):
    for _ in range(5):
        fn()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    return float(np.median(t)) * 1e3, float(t.min()) * 1e3


def workloads():
    l0 = bd.load_level0(("base",), "train", 1)
    yield "C2 4096 x one L0 puzzle", l0, 4096, np.zeros(4096, np.int64)
    paths = bench.level1_paths()
    B = 65536
    yield "C3 65536 Level-1", [PushWorldPuzzle(p) for p in paths], B, (np.arange(B, dtype=np.int64) * len(paths)) // B
    texts = list(bd.level0_texts().values())
    n_l0 = len(texts)
    for lv in (1, 2, 3, 4):
        for p in bd.level_paths(lv):
            with open(p) as f:
                texts.append(f.read())
    ids = np.sort(shard_puzzle_ids(c4_global_puzzle_ids(8 * B, n_l0, len(texts) - n_l0, 100), 0, 8))
    yield "C4 shard 14223 puzzles", _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0), B, ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=60)
    args = ap.parse_args()
    print("library:", os.path.basename(_capi.LIB_PATH))
    print("%-28s %-10s %14s %14s %16s %16s" % ("workload", "tables", "step us med", "step us min", "rollout64 us", "rollout steps/s"))
    for name, pool, B, ids in workloads():
        for lds in (2, 1):  # 2 = never (global rows through L1), 1 = always LDS
            vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True)
            try:
                vec.engine.set_option("step_lds_tables", lds)
            except ValueError as exc:
                print("%-28s %-10s %s" % (name, "lds", exc))
                continue
            vec.reset()
            g = torch.Generator(device=vec.device).manual_seed(1)
            acts = torch.randint(0, 4, (64, B), generator=g, device=vec.device, dtype=torch.uint8)
            it = [0]

            def one():
                vec.step(acts[it[0] % 64])
                it[0] += 1

            med, mn = timed(one, args.reps)
            rmed, _ = timed(lambda: vec.rollout(acts), max(10, args.reps // 4))
            print("%-28s %-10s %14.2f %14.2f %16.1f %16.3e" % (name, "lds" if lds == 1 else "global", med, mn, rmed, 64 * B / (rmed * 1e-6)),
                  flush=True)
            del vec


if __name__ == "__main__":
    main()
