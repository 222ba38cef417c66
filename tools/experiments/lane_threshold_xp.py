#!/usr/bin/env python3
"""Round-4 probe: one lane per environment (pw_step_lane_kernel / pw_rollout_lane_kernel) against the lane groups with the
per-workgroup choice of lanes (pw_step_group_mixed_kernel) by batch size, state only, on the C3 set (N_pad 16) and a C4-like
pool (N_pad 32): where the automatic switch (PW_OPT_STEP_LANE_BATCH) should sit."""
import glob
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pushworld_amd.config import BENCHMARK_PUZZLES_PATH  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402


def rates(pool, B, opts):
    ids = np.sort(np.arange(B) % len(pool))
    env = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True, tune=False, engine_options=opts)
    env.reset()
    T = 64
    acts = torch.randint(0, 4, (T, B), dtype=torch.uint8, device=env.device)
    env.rollout(acts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        env.rollout(acts)
    e1.record()
    e1.synchronize()
    r = 4 * T * B / (e0.elapsed_time(e1) * 1e-3)
    a1 = acts[0]
    for _ in range(20):
        env.step(a1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        env.step(a1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 300
    lane_from = env.engine.get_option("step_lane_batch")
    del env
    torch.cuda.empty_cache()
    return r, dt, lane_from


def main():
    l1 = [PushWorldPuzzle(f) for f in sorted(glob.glob(os.path.join(BENCHMARK_PUZZLES_PATH, "level1", "*.pwp")))]
    allp = [PushWorldPuzzle(f) for lv in ("level1", "level2", "level3", "level4")
            for f in sorted(glob.glob(os.path.join(BENCHMARK_PUZZLES_PATH, lv, "*.pwp")))]
    for name, pool in (("C3 set (N_pad 16)", l1), ("levels 1-4 (N_pad 32)", allp)):
        for B in (65536, 131072, 262144, 524288, 1048576):
            out = []
            for label, opts in (("groups", {"step_lane_batch": "never"}), ("lanes", {"step_kernel": "lane"}), ("automatic", {})):
                r, dt, lane_from = rates(pool, B, opts)
                out.append(f"{label}: rollouts {r:.3e}/s step {dt * 1e6:6.1f} us" + (f" (lanes from {lane_from})" if label == "automatic" else ""))
            print(f"{name:22s} B={B:8d}  " + "   ".join(out), flush=True)


if __name__ == "__main__":
    main()
