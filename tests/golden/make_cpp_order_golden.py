#!/usr/bin/env python3
"""C++-ORDER expectations for puzzles with MANY goal / movable ids, generated FROM THE PYTHON REFERENCE.

The reference's C++ engine cannot be built here (Boost), so the C++ object order -- ``pushworld_puzzle.cc:262-321``: the agent,
then the movables of the goals in the order of a ``std::map<std::string, ...>`` over the element ids (goal ids ascending BY
STRING: g1 < g10 < g11 < g2), then the remaining movables in the same string order (m10 < m12 < m2) -- was pinned only by the one
literal order of ``file_parsing.pwp`` (three movables).  Here: random puzzles with 10-13 movables whose ids run past 10, stepped by
the PYTHON reference (``puzzle.py:348-394``); its states are rearranged into the C++ order by the permutation that rule defines
(the rule is restated in ``cpp_names`` below from the C++ source, and ``python_names`` from ``puzzle.py:130-257``; the latter is
checked against the reference's own ``initial_state`` puzzle by puzzle), and stored with the 4 successors + moved-object masks
of sampled states for ``pw_expand4``.  Runs only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_cpp_order_golden.py     -> tests/golden/golden_cpp_order.json
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/python3/src")

import numpy as np  # noqa: E402
from pushworld.puzzle import PushWorldPuzzle  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))
N_PUZZLES = 24
N_STEPS = 300


def random_text(rng):
    """10-13 movables with ids from 1 .. 25 (most of them beyond 9), goals for about half of them, in a 16 x 14 grid."""
    cols, rows = 16, 14
    grid = [[set() for _ in range(cols)] for _ in range(rows)]

    def blob(n):
        x, y = int(rng.integers(0, cols)), int(rng.integers(0, rows))
        cells = [(x, y)]
        for _ in range(n - 1):
            bx, by = cells[int(rng.integers(0, len(cells)))]
            dx, dy = [(1, 0), (-1, 0), (0, 1), (0, -1)][int(rng.integers(0, 4))]
            nx, ny = bx + dx, by + dy
            if 0 <= nx < cols and 0 <= ny < rows and (nx, ny) not in cells:
                cells.append((nx, ny))
        return cells

    for _ in range(int(rng.integers(0, 10))):
        x, y = int(rng.integers(0, cols)), int(rng.integers(0, rows))
        grid[y][x].add("W")
    while True:
        cells = blob(int(rng.integers(1, 3)))
        if all(not grid[y][x] for x, y in cells):
            for x, y in cells:
                grid[y][x].add("A")
            break
    ids = [int(v) for v in rng.choice(np.arange(1, 26), size=int(rng.integers(9, 13)), replace=False)]
    if not any(v >= 10 for v in ids) or not any(v < 10 for v in ids):
        ids[0], ids[1] = 2, 10
    goal_ids = set(v for v in ids if rng.random() < 0.55)
    for k in ids:
        for _ in range(60):
            cells = blob(int(rng.integers(1, 4)))
            if all(not grid[y][x] for x, y in cells):
                for x, y in cells:
                    grid[y][x].add(f"M{k}")
                if k in goal_ids:
                    for (x, y) in blob(int(rng.integers(1, 3))):
                        if not any(t.startswith("G") for t in grid[y][x]):
                            grid[y][x].add(f"G{k}")
                break
    has_m = {t[1:] for row in grid for c in row for t in c if t.startswith("M")}
    lines = []
    for row in grid:
        toks = []
        for c in row:
            c = {t for t in c if not (t.startswith("G") and t[1:] not in has_m)}
            toks.append("+".join(sorted(c)) if c else ".")
        lines.append(" ".join(f"{t:>7s}" for t in toks))
    return "\n".join(lines) + "\n"


def element_ids_in_file_order(text):
    """ids (lower case) in the order the reference's parser first meets them: rows top to bottom, cells left to right, the parts
    of a cell in the order written (puzzle.py:133-149)."""
    seen = []
    for line in text.splitlines():
        for cell in line.split():
            for e in cell.split("+"):
                e = e.lower()
                if e != "." and e not in seen:
                    seen.append(e)
    return seen


def python_names(text):
    """puzzle.py:170-257: the agent, the movables of the goals in DESCENDING string order of the goal ids (the element ids are
    walked in ``sorted(reverse=True)`` order, :177-181), the other movables in the order they were first met in the file."""
    ids = element_ids_in_file_order(text)
    movables = ["a"]
    for e in sorted(ids, reverse=True):
        if e[0] == "g":
            movables.append("m" + e[1:])
    for e in ids:
        if e[0] == "m" and e not in movables:
            movables.append(e)
    return movables


def cpp_names(text):
    """pushworld_puzzle.cc:262-321: iteration over a std::map<std::string, PointSet> = ascending byte-wise string order."""
    ids = sorted(element_ids_in_file_order(text))
    objects = ["a"]
    goals = [e for e in ids if e[0] == "g"]
    for g in goals:
        objects.append("m" + g[1:])
    for e in ids:
        if e[0] == "m" and e not in objects:
            objects.append(e)
    return objects, goals


def main():
    rng = np.random.default_rng(20260929)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        n = 0
        while n < N_PUZZLES:
            text = random_text(rng)
            path = os.path.join(tmp, f"c{n}.pwp")
            with open(path, "w") as f:
                f.write(text)
            try:
                pz = PushWorldPuzzle(path)
            except Exception:  # noqa: BLE001  (e.g. a goal whose movable found no room)
                continue
            py = python_names(text)
            cpp, goals = cpp_names(text)
            if len(py) < 10 or sorted(py) != sorted(cpp) or py == cpp:
                continue
            # the restated Python order against the reference itself: top-left corner of every object's cells, + the border
            pos = {}
            for y, line in enumerate(text.splitlines()):
                for x, cell in enumerate(line.split()):
                    for e in cell.lower().split("+"):
                        if e != ".":
                            px, py_ = pos.get(e, (10**9, 10**9))
                            pos[e] = (min(px, x + 1), min(py_, y + 1))
            assert tuple(pos[m] for m in py) == tuple(tuple(int(v) for v in p) for p in pz.initial_state), "python_names is not the reference's order"
            perm = [py.index(m) for m in cpp]  # C++ index -> Python index
            assert tuple(pos["g" + m[1:]] for m in cpp[1:1 + len(goals)]) == tuple(pos[g] for g in goals)
            actions = [int(a) for a in rng.integers(0, 4, size=N_STEPS)]
            state = pz.initial_state
            states = [[list(map(int, state[i])) for i in perm]]
            goal_flags = [bool(pz.is_goal_state(state))]
            samples = []
            for t, a in enumerate(actions):
                if t % 15 == 0:  # the four successors of this state, C++ order, + which objects moved (pushworld_puzzle.cc:446-457)
                    succ, moved, goal = [], [], []
                    for b in range(4):
                        nxt = pz.get_next_state(state, b)
                        succ.append([list(map(int, nxt[i])) for i in perm])
                        moved.append(sum(1 << ci for ci, pi in enumerate(perm) if tuple(nxt[pi]) != tuple(state[pi])))
                        goal.append(bool(pz.is_goal_state(nxt)))
                    samples.append({"t": t, "succ": succ, "moved": moved, "goal": goal})
                state = pz.get_next_state(state, a)
                states.append([list(map(int, state[i])) for i in perm])
                goal_flags.append(bool(pz.is_goal_state(state)))
            out[f"cpporder:{n}"] = {"text": text, "python_names": py, "cpp_names": cpp, "cpp_goal_ids": goals,
                                    "goal_state_cpp": [list(pos[g]) for g in goals], "actions": actions, "states_cpp": states,
                                    "goal_flags": goal_flags, "expand": samples}
            n += 1
    with open(os.path.join(HERE, "golden_cpp_order.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    moved_any = sum(1 for v in out.values() for s in v["expand"] for m in s["moved"] if bin(m).count("1") > 1)
    print(len(out), "puzzles;", sum(len(v["cpp_names"]) for v in out.values()) / len(out), "objects on average;", moved_any, "sampled successors with pushes")


if __name__ == "__main__":
    main()
