#!/usr/bin/env python3
"""Small drivers for rocprofv3 passes over the kernels that are not the C3 render (tools/profile_hotpath.py is that one):

    rocprofv3 --kernel-trace --stats -d out -o name -- python tools/profile_kernels.py --what c4_step
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o name -- python tools/profile_kernels.py --what expand --puzzle "level4/Four Pistons.pwp"

  c4_step   rank 0's shard of config C4, state only: `--steps` single-step launches (pw_step_group_mixed_kernel) and 4 rollouts
  c2_step   config C2: 4 096 copies of one Level-0 puzzle, state only (bound: pw_step_seg_kernel)
  expand    pw_expand4 on the first `--states` states of a breadth-first search of `--puzzle` (repeated when the state space
            is smaller), `--steps` launches cycling through buffer sets beyond the Infinity Cache
  search    the GPU breadth-first search of `--puzzle` up to `--states` states (pw_search_* kernels)
  batch     pw_search_batch over `--states` generated Level-0 puzzles
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", required=True, choices=("c4_step", "c2_step", "expand", "search", "batch"))
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--states", type=int, default=4_000_000)
    ap.add_argument("--puzzle", default="level4/Four Pistons.pwp")
    ap.add_argument("--lds-tables", default="auto", choices=("auto", "never"))
    ap.add_argument("--lanes", action="store_true", help="c4_step: one lane per environment whatever the batch (PW_OPT_STEP_LANE_BATCH 1)")
    ap.add_argument("--rollouts", type=int, default=4, help="c4_step / c2_step: 64-step pw_rollout launches after the single steps")
    args = ap.parse_args()
    from tools import config_suite as cs

    if args.what == "c4_step":
        import bench

        ns = argparse.Namespace(envs_per_gpu=65536, obs="none", config="c4", max_steps=200, bw=1, ppc=3, tune_allocations=None)
        vec = bench.build_workload(ns, 0, 8, 0)["vec"]
        if args.lanes:
            vec.engine.set_option("step_lane_batch", 1)
        vec.reset()
        acts = cs.actions_for(64, 65536, vec.device, 100)
        torch.cuda.synchronize()
        for t in range(args.steps):
            vec.step(acts[t % 64])
        for _ in range(args.rollouts):
            vec.rollout(acts)
        torch.cuda.synchronize()
        print("c4_step done", args.steps, vec.counters())
    elif args.what == "c2_step":
        from pushworld_amd import benchmark_data as bd
        from pushworld_amd.puzzle import PushWorldPuzzle
        from pushworld_amd.vec_env import VecPushWorld

        text = next(iter(bd.level0_texts(("base",), "train", 1).values()))
        vec = VecPushWorld([PushWorldPuzzle(text=text)], 4096, max_steps=100, observation=None, autoreset=True)
        vec.reset()
        acts = cs.actions_for(64, 4096, vec.device, 0)
        for t in range(args.steps):
            vec.step(acts[t % 64])
        for _ in range(args.rollouts):
            vec.rollout(acts)
        torch.cuda.synchronize()
        print("c2_step done", vec.counters())
    elif args.what == "expand":
        pz, st_host, exhausted, distinct = cs.c5_frontier(args.puzzle, args.states)
        F, N = st_host.shape
        eng = pz._engine()
        eng.set_option("expand_lds_tables", args.lds_tables)
        dev = eng.device
        nbuf = max(1, -(-cs.CACHE_BUST_BYTES // (F * (20 * N + 20))))
        sets = [(torch.as_tensor(st_host).to(dev), torch.empty((F, 4, N), dtype=torch.int32, device=dev),
                 torch.empty((F, 4), dtype=torch.int32, device=dev), torch.empty((F, 4), dtype=torch.uint8, device=dev))
                for _ in range(nbuf)]
        torch.cuda.synchronize()
        for t in range(args.steps):
            s = sets[t % nbuf]
            eng.expand4(0, s[0], s[1], s[2], s[3])
        torch.cuda.synchronize()
        print("expand done", args.puzzle, "F", F, "N", N, "distinct", distinct, "buffer sets", nbuf, "algorithmic bytes per launch", F * (20 * N + 20))
    elif args.what == "search":
        from pushworld_amd.config import BENCHMARK_PUZZLES_PATH
        from pushworld_amd.puzzle import PushWorldPuzzle
        from pushworld_amd.search import BreadthFirstSearch

        pz = PushWorldPuzzle(os.path.join(BENCHMARK_PUZZLES_PATH, args.puzzle), order="cpp")
        bfs = BreadthFirstSearch(pz, max_states=args.states + (args.states >> 1))
        bfs.begin()
        try:
            while bfs.total_states < args.states and not bfs.exhausted:
                bfs.expand()
        except ValueError:
            pass
        print("search done", args.puzzle, "states", bfs.total_states, "layers", len(bfs.layers))
        bfs.close()
    else:
        from pushworld_amd import _capi
        from pushworld_amd.generate import generate_level0_set
        from pushworld_amd.search import search_batch

        pset, _, _ = generate_level0_set(min(args.states, 65536), device=0, random_seed=7)
        eng = _capi.Engine(pset, None, 3, 1, _capi.OBS_U8)
        for _ in range(max(1, args.steps // 4)):
            v, pl, ns = search_batch(eng, None, max_states=1 << 16)
        print("batch done", len(pset), "puzzles; solved", int((v == 1).sum()), "unsolvable", int((v == 0).sum()), "unknown", int((v == 2).sum()),
              "states", int(ns.sum()))


if __name__ == "__main__":
    main()
