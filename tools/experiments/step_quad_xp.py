#!/usr/bin/env python3
"""Round-5 experiment: the C4 shard / the C3 set / the Level-0 half alone / the Level 1-4 half alone, state only, with the
16 x 16 whole-grid formulation (PW_OPT_STEP_QUAD16 auto) and without (never): one step per launch (HIP events around 200
launches, and the library's own per-launch events) and 64-step rollouts.

    python tools/experiments/step_quad_xp.py [--sets c4,c3,l0,hi]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(kind, B=65536):
    import bench
    from pushworld_amd import _capi
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    if kind == "c4":
        ns = argparse.Namespace(envs_per_gpu=B, obs="none", config="c4", max_steps=200, bw=1, ppc=3, tune_allocations=None)
        return bench.build_workload(ns, 0, 8, 0)["vec"]
    if kind == "c3":
        texts = [open(p).read() for p in bench.level1_paths()]
        ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
        return VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True)
    l0 = list(bd.level0_texts().values())
    hi = [open(p).read() for lv in (1, 2, 3, 4) for p in bd.level_paths(lv)]
    texts = l0 + hi
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    rng = np.random.default_rng(100)
    if kind == "l0":
        ids = np.sort(rng.integers(0, len(l0), size=B))
    elif kind.startswith("hi") and len(kind) > 2:  # "hi8": only every (223 / 8)-th Level 1-4 puzzle -- a small table working set
        k = int(kind[2:])
        pick = np.arange(0, len(hi), max(1, len(hi) // k))[:k]
        ids = np.sort(len(l0) + pick[rng.integers(0, len(pick), size=B)])
    else:
        ids = np.sort(len(l0) + rng.integers(0, len(hi), size=B))
    return VecPushWorld(pset, B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True)


def measure(vec, label):
    from tools import config_suite as cs

    B = vec.num_envs
    acts = cs.actions_for(64, B, vec.device, 100)
    vec.reset()
    it = [0]

    def one():
        vec.step(acts[it[0] % 64])
        it[0] += 1

    for _ in range(50):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(200):
            one()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200)
    ms = cs.launch_ms(vec.engine, one, 300)
    vec.rollout(acts)
    torch.cuda.synchronize()
    rb = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(4):
            vec.rollout(acts)
        e1.record()
        torch.cuda.synchronize()
        rb = min(rb, e0.elapsed_time(e1) / 4)
    print(f"{label:28s} step back-to-back {1e3 * best:7.2f} us  ({B / best / 1e6:7.2f} e9/s)   per-launch events median {1e3 * np.median(ms):6.2f} min {1e3 * ms.min():6.2f} us"
          f"   rollout64 {1e3 * rb:7.1f} us ({64 * B / rb / 1e6:6.2f} e9/s)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="c4,c3,l0,hi")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--modes", default="auto:groups,never:groups,auto:lanes,auto:groups,never:groups,auto:lanes",
                    help="comma-separated quad16:kernel pairs (under rocprofv3: one pair per process, the kernel names do not say which)")
    args = ap.parse_args()
    for kind in args.sets.split(","):
        vec = build(kind, args.envs)
        print(kind, "N_pad", vec.num_objects_padded, "puzzles", vec.num_puzzles, "with a 16 x 16 record", vec.engine.get_option("step_quad16_puzzles"), flush=True)
        for mode in args.modes.split(","):
            quad, kern, *rest = mode.split(":")
            vec.engine.set_option("step_block_order", int(rest[0]) if rest else 0)
            lanes = kern == "lanes"
            vec.engine.set_option("step_quad16", quad)
            vec.engine.set_option("step_lane_batch", 1 if lanes else 2**31)
            measure(vec, f"{kind} quad16={quad} {'lanes' if lanes else 'groups'} order={rest[0] if rest else 0}")
        del vec
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
