#!/usr/bin/env python3
"""Round-5 experiment: the state-only single step of 65 536 environments of ONE Level 1-4 puzzle at a time (tables hot, every
workgroup the same puzzle) -- which puzzles set the pace of the mixed launches.  One line per puzzle, sorted by time.
    python tools/experiments/step_per_puzzle.py [--envs 65536] > gpurun_out/step_per_puzzle.txt"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    args = ap.parse_args()
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld
    from tools import config_suite as cs

    B = args.envs
    rows = []
    acts = cs.actions_for(64, B, torch.device("cuda", 0), 100)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for lv in (1, 2, 3, 4):
        for path in bd.level_paths(lv):
            vec = VecPushWorld([PushWorldPuzzle(path)], B, max_steps=200, observation=None, autoreset=True, device=0)
            vec.reset()
            for k in range(100):
                vec.step(acts[k % 64])
            best = 1e9
            for _ in range(3):
                e0.record()
                for k in range(200):
                    vec.step(acts[k % 64])
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 200)
            rows.append((1e3 * best, os.path.basename(path), vec.num_objects_padded, vec.engine.get_option("step_quad16_puzzles")))
            del vec
    rows.sort()
    for us, name, npad, quad in rows:
        print(f"{us:7.2f} us  N_pad {npad:2d}  quad16 {quad}  {name}")
    t = np.array([r[0] for r in rows])
    print(f"# {len(rows)} puzzles: min / median / p90 / max = {t.min():.2f} / {np.median(t):.2f} / {np.quantile(t, 0.9):.2f} / {t.max():.2f} us")


if __name__ == "__main__":
    main()
