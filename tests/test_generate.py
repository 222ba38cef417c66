"""pushworld_amd.generate: the level-0 recipe of python3/src/pushworld/generate.py (structure checks on
CPU) and the GPU solvability filter against the oracle's breadth-first search."""
import os
from collections import deque

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cells(text):
    out = {}
    for y, line in enumerate(text.splitlines()):
        for x, tok in enumerate(line.split()):
            if tok != ".":
                out.setdefault(tok, set()).add((x, y))
    return out


def _norm(cells):
    mx, my = min(x for x, _ in cells), min(y for _, y in cells)
    return sorted((x - mx, y - my) for x, y in cells)


def test_generated_puzzles_follow_the_recipe():
    from pushworld_amd.generate import COMPLEX_SHAPES, FailedToGenerateError, generate_puzzle

    rng = np.random.default_rng(5)
    shapes = [sorted((c, r) for r, c in s) for s in COMPLEX_SHAPES]
    n_two = 0
    for k in range(300):
        w, h = int(rng.integers(8, 13)), int(rng.integers(8, 13))
        walls, obst, goals = int(rng.integers(2, 5)), int(rng.integers(1, 3)), int(rng.integers(1, 3))
        try:
            text = generate_puzzle(w, h, walls, obst, goals, COMPLEX_SHAPES, rng)
        except FailedToGenerateError:
            continue
        rows = text.splitlines()
        assert len(rows) == h and all(len(r.split()) == w for r in rows)
        c = _cells(text)
        assert set(c) == {"A", "W"} | {f"M{i}" for i in range(1, 1 + goals + obst)} | {f"G{i}" for i in range(1, 1 + goals)}
        assert len(c["W"]) == walls
        assert _norm(c["M1"]) == _norm(c["G1"]) and _norm(c["M1"]) in shapes
        if goals == 2:
            n_two += 1
            assert _norm(c["M2"]) == _norm(c["G2"]) != _norm(c["M1"])
        for name, cells in c.items():
            if name != "W":
                assert _norm(cells) in shapes, name
        assert sum(len(v) for v in c.values()) == len(set().union(*c.values()))  # nothing overlaps
    assert n_two > 50
    a = generate_puzzle(9, 9, 3, 2, 1, COMPLEX_SHAPES, np.random.default_rng(1))
    assert a == generate_puzzle(9, 9, 3, 2, 1, COMPLEX_SHAPES, np.random.default_rng(1))  # reproducible per seed
    with pytest.raises(FailedToGenerateError):
        generate_puzzle(2, 2, 4, 2, 1, COMPLEX_SHAPES, np.random.default_rng(0))


def test_generate_level0_argument_checks(tmp_path):
    from pushworld_amd.generate import generate_level0_puzzles

    d = str(tmp_path / "out")
    for kw in (dict(num_puzzles=0), dict(min_puzzle_size=1), dict(min_puzzle_size=9, max_puzzle_size=8), dict(min_num_walls=-1),
               dict(min_num_obstacles=3, max_num_obstacles=2), dict(min_num_goal_objects=0), dict(max_num_goal_objects=3),
               dict(object_shapes="round")):
        with pytest.raises(ValueError):
            generate_level0_puzzles(d, filter_puzzles=False, **kw)
    generate_level0_puzzles(d, num_puzzles=7, random_seed=3, filter_puzzles=False)
    assert sorted(os.listdir(d)) == [f"puzzle_{i}.pwp" for i in range(7)]
    with pytest.raises(ValueError):
        generate_level0_puzzles(d, num_puzzles=1, filter_puzzles=False)  # not empty


@pytest.mark.gpu
def test_solvability_filter_is_exact(tmp_path):
    """Every kept puzzle comes with a valid plan, every dropped one is proven unsolvable by the host
    breadth-first search over the oracle (small boards, so both searches finish)."""
    from oracle import c_oracle
    from pushworld_amd.generate import generate_level0_puzzles, solve
    from pushworld_amd.puzzle import PushWorldPuzzle

    d = str(tmp_path / "lvl")
    n = 30
    generate_level0_puzzles(d, num_puzzles=n, random_seed=11, filter_puzzles=False, min_puzzle_size=5, max_puzzle_size=6,
                            min_num_walls=3, max_num_walls=6, min_num_obstacles=1, max_num_obstacles=2)
    texts = [open(os.path.join(d, f"puzzle_{i}.pwp")).read() for i in range(n)]

    def host_solvable(text):
        oz = c_oracle.COraclePuzzle(text)
        seen, q = {oz.initial_state}, deque([oz.initial_state])
        while q:
            s = q.popleft()
            if oz.py.is_goal_state(s):
                return True
            for a in range(4):
                t = oz.get_next_state(s, a)
                if t not in seen:
                    seen.add(t)
                    q.append(t)
            assert len(seen) < 400000
        return False

    verdicts = []
    for text in texts:
        plan, verdict = solve(text, max_states=500000)
        assert verdict in ("solved", "unsolvable")
        verdicts.append(verdict)
        if plan is not None:
            assert PushWorldPuzzle(text=text).is_valid_plan(plan)
        assert host_solvable(text) == (verdict == "solved")
    assert "solved" in verdicts and "unsolvable" in verdicts
    from pushworld_amd.generate import filter_puzzles_by_solvability

    kept = filter_puzzles_by_solvability(d, None, n, max_states=500000)
    assert kept == verdicts.count("solved")
    assert sorted(os.listdir(d)) == sorted(f"puzzle_{i}.pwp" for i in range(kept))
    survivors = [t for t, v in zip(texts, verdicts) if v == "solved"]
    assert [open(os.path.join(d, f"puzzle_{i}.pwp")).read() for i in range(kept)] == survivors
