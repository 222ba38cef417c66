#!/usr/bin/env python3
"""Runs the GPU planner over the benchmark puzzles: IW(1), then IW(2), then plain breadth-first search with a
state cap -- all three are pw_search_* (closed set, frontier and novelty tables in HBM).  Prints one line per
puzzle and a summary; plans are validated with PushWorldPuzzle.is_valid_plan (GPU step engine).

    python tools/solve_benchmark.py [--levels 1 2 3 4] [--max-states N]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402,F401


def human_plan_length(path):
    sol = path.replace(os.sep + "puzzles" + os.sep, os.sep + "solutions" + os.sep)[:-4] + ".yaml"
    if not os.path.exists(sol):
        return None
    with open(sol) as f:
        for line in f:
            if line.startswith("plan:"):
                return len(line.split(":", 1)[1].strip())
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--max-states", type=int, default=30_000_000)
    args = ap.parse_args()
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch

    rows = []
    t_all = time.perf_counter()
    for lv in args.levels:
        for path in bd.level_paths(lv):
            pz = PushWorldPuzzle(path)
            name = f"level{lv}/{os.path.basename(path)[:-4]}"
            res = {"name": name, "N": pz.num_movables, "human": human_plan_length(path), "by": None, "plan": None,
                   "states": 0, "seconds": 0.0}
            for label, width, cap in (("IW(1)", 1, 1 << 20), ("IW(2)", 2, args.max_states), ("BFS", 0, args.max_states)):
                t0 = time.perf_counter()
                try:
                    bfs = BreadthFirstSearch(pz, max_states=cap, novelty_width=width)
                except MemoryError:
                    continue
                try:
                    plan = bfs.solve()
                except ValueError:  # store full
                    plan = None
                dt = time.perf_counter() - t0
                res["seconds"] += dt
                res["states"] = bfs.total_states
                bfs.close()
                if plan is not None:
                    assert pz.is_valid_plan(plan), name
                    res.update(by=label, plan=len(plan))
                    break
            rows.append(res)
            print(f"{name:45s} N={res['N']:2d} human={res['human']!s:>4s} solved_by={res['by']!s:6s} plan={res['plan']!s:>4s} "
                  f"states={res['states']:>10d} {res['seconds'] * 1e3:9.1f} ms", flush=True)
    total = time.perf_counter() - t_all
    print()
    for lv in args.levels:
        sub = [r for r in rows if r["name"].startswith(f"level{lv}/")]
        by = {k: sum(1 for r in sub if r["by"] == k) for k in ("IW(1)", "IW(2)", "BFS", None)}
        print(f"level {lv}: {len(sub)} puzzles, solved {len(sub) - by[None]} (IW(1) {by['IW(1)']}, IW(2) {by['IW(2)']}, BFS {by['BFS']}), "
              f"unsolved within caps {by[None]}")
    print(f"total wall time {total:.1f} s (includes puzzle parsing and table allocation)")


if __name__ == "__main__":
    main()
