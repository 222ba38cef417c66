#!/usr/bin/env python3
"""Distribution statistics of the REFERENCE's level-0 generator (python3/src/pushworld/generate.py:74-259), for
tests/test_generate_stats.py.  Runs in the build container only (imports the reference); writes
tests/golden/golden_generator_stats.json.

Bit equality with the reference is undefined (its stream is Python's Mersenne Twister; ours is counter based), so
the pin is distributional: per configuration ~20 000 puzzles of generate_level0_puzzles(filter_puzzles=False) ->
histograms of width, height, wall / obstacle / goal-object counts, the shape index of every role, and the share of
generate_puzzle attempts that raised FailedToGenerateError.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_generator_stats.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/python3/src")
sys.dont_write_bytecode = True

from pushworld import generate as ref  # noqa: E402

N = 20000
CONFIGS = {  # the configurations of tests/test_generate_grids.py
    "default": dict(),
    "dense": dict(min_puzzle_size=5, max_puzzle_size=7, min_num_walls=0, max_num_walls=9, min_num_obstacles=0,
                  max_num_obstacles=5, min_num_goal_objects=1, max_num_goal_objects=2),
    "simple": dict(object_shapes="simple", min_puzzle_size=3, max_puzzle_size=4, max_num_walls=2),
}
SHAPES = [  # (row, column) offsets, generate.py:215-225
    [(0, 0)], [(0, 0), (0, 1)], [(0, 0), (1, 0)], [(0, 0), (1, 0), (1, 1)], [(0, 0), (0, 1), (1, 1)],
    [(0, 0), (0, 1), (1, 0)], [(1, 0), (0, 1), (1, 1)], [(0, 0), (0, 1), (0, 2)], [(0, 0), (1, 0), (2, 0)],
]
SHAPE_INDEX = {frozenset(s): i for i, s in enumerate(SHAPES)}


def puzzle_stats(text):
    rows = [r.split() for r in text.split("\n")]
    cells = {}
    for y, row in enumerate(rows):
        for x, tok in enumerate(row):
            if tok != ".":
                cells.setdefault(tok, []).append((y, x))

    def shape_of(name):
        c = cells[name]
        y0, x0 = min(y for y, _ in c), min(x for _, x in c)
        return SHAPE_INDEX[frozenset((y - y0, x - x0) for y, x in c)]

    movers = sorted(int(k[1:]) for k in cells if k[0] == "M")
    goals = sorted(int(k[1:]) for k in cells if k[0] == "G")
    n_goals = len(goals)
    out = {"width": len(rows[0]), "height": len(rows), "walls": len(cells.get("W", ())), "goals": n_goals,
           "obstacles": len(movers) - n_goals, "shape_m1": shape_of("M1"), "shape_agent": shape_of("A"),
           "shape_obstacles": [shape_of("M%d" % k) for k in movers if k > n_goals]}
    if n_goals == 2:
        out["shape_m2"] = shape_of("M2")
    return out


def hist(values, lo, hi):
    h = [0] * (hi - lo + 1)
    for v in values:
        h[v - lo] += 1
    return h


def main():
    result = {"_n": N, "_shapes": SHAPES}
    for name, kw in CONFIGS.items():
        failures = [0]
        orig = ref.generate_puzzle

        def counted(*a, **k):
            try:
                return orig(*a, **k)
            except ref.FailedToGenerateError:
                failures[0] += 1
                raise

        ref.generate_puzzle = counted
        with tempfile.TemporaryDirectory() as d:
            ref.generate_level0_puzzles(os.path.join(d, "out"), num_puzzles=N, random_seed=20260928, filter_puzzles=False, **kw)
            stats = []
            for i in range(N):
                with open(os.path.join(d, "out", "puzzle_%d.pwp" % i)) as f:
                    stats.append(puzzle_stats(f.read()))
        ref.generate_puzzle = orig
        lo, hi = kw.get("min_puzzle_size", 8), kw.get("max_puzzle_size", 12)
        n_shapes = 1 if kw.get("object_shapes") == "simple" else 9
        result[name] = {
            "kwargs": kw,
            "width": hist([s["width"] for s in stats], lo, hi), "height": hist([s["height"] for s in stats], lo, hi),
            "walls": hist([s["walls"] for s in stats], kw.get("min_num_walls", 2), kw.get("max_num_walls", 4)),
            "obstacles": hist([s["obstacles"] for s in stats], kw.get("min_num_obstacles", 1), kw.get("max_num_obstacles", 2)),
            "goals": hist([s["goals"] for s in stats], kw.get("min_num_goal_objects", 1), kw.get("max_num_goal_objects", 1)),
            "shape_m1": hist([s["shape_m1"] for s in stats], 0, n_shapes - 1),
            "shape_m2": hist([s["shape_m2"] for s in stats if "shape_m2" in s], 0, n_shapes - 1),
            "shape_agent": hist([s["shape_agent"] for s in stats], 0, n_shapes - 1),
            "shape_obstacles": hist([v for s in stats for v in s["shape_obstacles"]], 0, n_shapes - 1),
            "failed_attempts": failures[0], "attempts": failures[0] + N,
        }
        print(name, json.dumps(result[name]), flush=True)
    with open(os.path.join(HERE, "golden_generator_stats.json"), "w") as f:
        json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
