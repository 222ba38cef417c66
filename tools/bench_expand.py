#!/usr/bin/env python3
"""Config C5: successor-expansion throughput (parents/s) on BFS frontiers of benchmark puzzles."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pushworld_amd.config import BENCHMARK_PUZZLES_PATH  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402


def frontier(pz, target):
    init = np.array([[x * 10000 + y for (x, y) in pz.initial_state]], dtype=np.int32)
    seen = {tuple(init[0])}
    layer = init
    allstates = [init]
    while sum(len(a) for a in allstates) < target and len(layer):
        succ, _, _ = pz.expand4(layer)
        rows = np.unique(succ.cpu().numpy().reshape(-1, succ.shape[-1]), axis=0)
        new = [r for r in map(tuple, rows) if r not in seen]
        seen.update(new)
        layer = np.array(new, dtype=np.int32).reshape(-1, init.shape[1])
        allstates.append(layer)
    return np.concatenate(allstates)


def main():
    # kernels: "groups" = one lane group per state whatever the frontier size (PW_OPT_STEP_LANE_BATCH never), "auto" = the
    # default (one lane per state from 131 072 states on, puzzles with tables and <= 16 movables), "lanes" = always
    sizes = [int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(",")] if "--sizes" in sys.argv else [1_000_000]
    cases = [(rel, 200000, "auto", k, F) for rel in ("level1/2 Obstacle.pwp", "level2/Pull Dont Push.pwp", "level4/Four Pistons.pwp")
             for F in sizes for k in ("groups", "auto")]
    cases += [("level4/Mind The Gap.pwp", 200000, "none", "auto", sizes[-1]), ("level4/Mind The Gap.pwp", 200000, "auto", "groups", sizes[-1]),
              ("level4/Mind The Gap.pwp", 200000, "auto", "auto", sizes[-1])]
    fronts = {}
    for rel, target, mode, kernel, size in cases:
        pz = PushWorldPuzzle(os.path.join(BENCHMARK_PUZZLES_PATH, rel), order="cpp")
        pz._engine().set_option("step_tables", mode)  # PW_OPT_STEP_TABLES
        if kernel == "groups":
            pz._engine().set_option("step_lane_batch", "never")
        elif kernel == "lanes":
            pz._engine().set_option("step_kernel", "lane")
        t0 = time.time()
        if rel not in fronts:
            fronts[rel] = frontier(pz, target)
        st = fronts[rel]
        rel = "%s [tables %s, %s]" % (rel, mode, kernel)
        reps = max(1, -(-size // len(st)))
        states = torch.as_tensor(np.tile(st, (reps, 1))[:max(size, 1)]).to("cuda:0")
        F, N = states.shape
        eng = pz._engine()
        succ = torch.empty((F, 4, N), dtype=torch.int32, device="cuda:0")
        moved = torch.empty((F, 4), dtype=torch.int32, device="cuda:0")
        goal = torch.empty((F, 4), dtype=torch.uint8, device="cuda:0")
        eng.expand4(0, states, succ, moved, goal)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.expand4(0, states, succ, moved, goal)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        algo = F * (20 * N + 20)
        print(f"{rel:52s} N={N:2d} distinct={len(st):7d} F={F:8d}  {ms:7.3f} ms  {F / ms * 1e3:11.3e} parents/s  "
              f"{4 * F / ms * 1e3:11.3e} successors/s  {algo / ms / 1e6:7.1f} GB/s algorithmic  (frontier build {time.time() - t0:.1f}s)",
              flush=True)


if __name__ == "__main__":
    main()
