#!/usr/bin/env python3
"""GPU breadth-first search throughput (pw_search_*): explores benchmark puzzles layer by layer until the goal,
exhaustion or --max-states, and reports parents expanded per second (each parent = 4 successors + closed-set
lookups + insertion) next to a host FIFO search over the oracle's C port on a bounded prefix.

    python tools/bench_search.py [--max-states N]
"""
import argparse
import json
import os
import sys
import time
from collections import deque

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def host_rate(text, seconds=3.0):
    from oracle import c_oracle

    oz = c_oracle.COraclePuzzle(text, order="cpp")
    s0 = oz.initial_state
    seen = {s0}
    q = deque([s0])
    t0 = time.perf_counter()
    n = 0
    while q and time.perf_counter() - t0 < seconds:
        s = q.popleft()
        n += 1
        for a in range(4):
            t = oz.get_next_state(s, a)
            if t != s and t not in seen:
                seen.add(t)
                q.append(t)
    return n / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-states", type=int, default=40_000_000)
    ap.add_argument("--groups", action="store_true", help="lane groups for every pass (PW_OPT_STEP_LANE_BATCH never) instead "
                    "of one lane per parent from 131 072 parents on")
    ap.add_argument("--keys", default="fingerprint", choices=("fingerprint", "exact"), help="closed set: fingerprint + index entries / exact 63-bit keys where they fit")
    ap.add_argument("--only", default=None, help="substring of the puzzle path")
    ap.add_argument("--stop-at", type=int, default=0, help="stop once this many states are in the store (a table sized for --max-states, a "
                    "shorter search: what the size of the closed set's table costs)")
    args = ap.parse_args()
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch

    d = os.path.join(ROOT, "pushworld_amd", "data", "puzzles")
    out = {}
    for rel in ("level1/2 Obstacle.pwp", "level1/Choose Wisely.pwp", "level2/Pull Dont Push.pwp", "level4/Four Pistons.pwp"):
        if args.only and args.only not in rel:
            continue
        with open(os.path.join(d, rel)) as f:
            text = f.read()
        pz = PushWorldPuzzle(text=text, order="cpp")
        pz._engine().set_option("search_keys", args.keys)
        if args.groups:
            pz._engine().set_option("step_lane_batch", "never")
        bfs = BreadthFirstSearch(pz, max_states=args.max_states)
        bfs.begin()
        bfs.expand()  # warm-up launch
        bfs.begin()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status = "exhausted"
        try:
            while not bfs.exhausted:
                info = bfs.expand()
                if info.goal_index >= 0:
                    status = "solved"
                    break
                if args.stop_at and bfs.total_states >= args.stop_at:
                    status = "stopped"
                    break
        except ValueError:
            status = "store full"
        dt = time.perf_counter() - t0
        # parents whose successors were generated: everything before the newest layer
        expanded = bfs.total_states if status == "exhausted" else bfs.layers[-1][0]
        ent = {"status": status, "movables": pz.num_movables, "depth": len(bfs.layers) - 1, "states": bfs.total_states,
               "parents_expanded": expanded, "seconds": dt, "parents_per_s": expanded / dt,
               "largest_layer": max(c for _, c in bfs.layers)}
        if status == "solved":
            plan = bfs.plan(bfs.goal_index)
            ent["plan_length"] = len(plan)
            ent["plan_valid"] = bool(pz.is_valid_plan(plan))
        ent["host_fifo_parents_per_s"] = host_rate(text)
        if bfs.total_states <= (1 << 18) and status != "store full":
            # the same search as ONE launch (pw_search_batch with n = 1: the whole loop inside the kernel)
            from pushworld_amd.search import shortest_plan

            shortest_plan(pz)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                plan1, v1 = shortest_plan(pz)
            dt1 = (time.perf_counter() - t1) / reps
            ent["single_launch"] = {"seconds": dt1, "parents_per_s": expanded / dt1, "verdict": int(v1),
                                    "plan_length": None if plan1 is None else len(plan1)}
        out[rel] = ent
        bfs.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
