"""Random level-0 style puzzles with an exact solvability filter on the GPU.

Same recipe and argument names as the reference generator (python3/src/pushworld/generate.py:28-259):
a goal object and its goal of one shape, optionally a second pair of a different shape, the agent,
obstacles, single-cell walls, each dropped at a uniformly random free position (100 attempts per
object, then the puzzle is abandoned).  Two deliberate differences:

* the random stream is ``numpy.random.default_rng(random_seed)``, not the interpreter-global
  ``random`` module -- puzzles are reproducible per seed here but are not the reference's puzzles;
* the filter does not run the RGD planner for ``time_limit`` seconds per puzzle
  (generate.py:262-297); it searches the state space on the GPU (``pushworld_amd.search``):
  width-limited search first, then breadth-first search until a plan is found, the space is
  exhausted (= proven unsolvable) or ``max_states`` / ``time_limit`` is hit (= dropped, like a
  planner time-out).
"""
from __future__ import annotations

import os
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .config import PUZZLE_EXTENSION

Shape = Sequence[Tuple[int, int]]  # (row offset, column offset) cells

SIMPLE_SHAPES: List[Shape] = [[(0, 0)]]
COMPLEX_SHAPES: List[Shape] = [  # the monomino, the dominoes and the trominoes, generate.py:215-225
    [(0, 0)],
    [(0, 0), (0, 1)],
    [(0, 0), (1, 0)],
    [(0, 0), (1, 0), (1, 1)],
    [(0, 0), (0, 1), (1, 1)],
    [(0, 0), (0, 1), (1, 0)],
    [(1, 0), (0, 1), (1, 1)],
    [(0, 0), (0, 1), (0, 2)],
    [(0, 0), (1, 0), (2, 0)],
]
MAX_PLACEMENT_ATTEMPTS = 100


class FailedToGenerateError(Exception):
    """An object found no free spot in ``MAX_PLACEMENT_ATTEMPTS`` tries (generate.py:23-25)."""


def _drop(grid: np.ndarray, symbols: List[List[str]], name: str, shape: Shape, rng: np.random.Generator) -> None:
    rows = 1 + max(r for r, _ in shape)
    cols = 1 + max(c for _, c in shape)
    h, w = grid.shape
    if cols > w or rows > h:  # (the reference fails with IndexError on random.choice of an empty range)
        raise FailedToGenerateError()
    for _ in range(MAX_PLACEMENT_ATTEMPTS):
        x = int(rng.integers(0, w + 1 - cols))
        y = int(rng.integers(0, h + 1 - rows))
        if all(not grid[y + r, x + c] for r, c in shape):
            for r, c in shape:
                grid[y + r, x + c] = True
                symbols[y + r][x + c] = name
            return
    raise FailedToGenerateError()


def generate_puzzle(puzzle_width: int, puzzle_height: int, num_walls: int, num_obstacles: int, num_goal_objects: int,
                    possible_object_shapes: Sequence[Shape], rng: Optional[np.random.Generator] = None) -> str:
    """One puzzle text (generate.py:74-133); raises ``FailedToGenerateError`` if an object does not fit."""
    assert len(possible_object_shapes) >= num_goal_objects, "need a distinct shape for each goal object"
    rng = np.random.default_rng() if rng is None else rng
    grid = np.zeros((puzzle_height, puzzle_width), dtype=bool)
    symbols = [["."] * puzzle_width for _ in range(puzzle_height)]

    def pick():
        return possible_object_shapes[int(rng.integers(0, len(possible_object_shapes)))]

    first = pick()
    _drop(grid, symbols, "M1", first, rng)
    _drop(grid, symbols, "G1", first, rng)
    if num_goal_objects == 2:
        second = pick()
        while list(second) == list(first):
            second = pick()
        _drop(grid, symbols, "M2", second, rng)
        _drop(grid, symbols, "G2", second, rng)
    _drop(grid, symbols, "A", pick(), rng)
    for i in range(num_obstacles):
        _drop(grid, symbols, f"M{1 + i + num_goal_objects}", pick(), rng)
    for _ in range(num_walls):
        _drop(grid, symbols, "W", [(0, 0)], rng)
    return "\n".join("  ".join(row) for row in symbols)


def solve(puzzle_text: str, max_states: int = 2_000_000, time_limit: Optional[float] = None):
    """``(plan, verdict)`` with verdict "solved" | "unsolvable" (space exhausted) | "unknown" (cap hit).
    IW(2) finds most plans in a few hundred states; breadth-first search decides the rest exactly."""
    from .puzzle import PushWorldPuzzle
    from .search import BreadthFirstSearch

    puzzle = PushWorldPuzzle(text=puzzle_text)
    t0 = time.perf_counter()
    for width in (2, 0):
        bfs = BreadthFirstSearch(puzzle, max_states=max_states, novelty_width=width)
        bfs.begin()
        full = False
        try:
            while bfs.goal_index < 0 and not bfs.exhausted:
                if time_limit is not None and time.perf_counter() - t0 > time_limit:
                    return None, "unknown"
                bfs.expand()
        except ValueError:  # store full: the last layer is incomplete, but a goal found in it is still a goal
            full = True
        finally:
            plan = bfs.plan(bfs.goal_index) if bfs.goal_index >= 0 else None
            bfs.close()
        if plan is not None:
            return plan, "solved"
        if width == 0:
            return None, ("unknown" if full else "unsolvable")
    return None, "unknown"


def filter_puzzles_by_solvability(path: str, time_limit: Optional[float], num_puzzles: int,
                                  max_states: int = 2_000_000) -> int:
    """Keeps the puzzles ``puzzle_<i>.pwp`` for which a plan is found and renumbers them densely
    (generate.py:262-297).  Returns the number kept."""
    kept = 0
    for i in range(num_puzzles):
        src = os.path.join(path, f"puzzle_{i}{PUZZLE_EXTENSION}")
        with open(src) as f:
            plan, _ = solve(f.read(), max_states=max_states, time_limit=time_limit)
        if plan is None:
            os.remove(src)
        else:
            if kept != i:
                os.rename(src, os.path.join(path, f"puzzle_{kept}{PUZZLE_EXTENSION}"))
            kept += 1
    print(f"{kept}/{num_puzzles} were solvable")
    return kept


def generate_level0_puzzles(save_location_path: str, num_puzzles: int = 5, random_seed: int = 0, filter_puzzles: bool = True,
                            time_limit: Optional[float] = 2, min_puzzle_size: int = 8, max_puzzle_size: int = 12,
                            min_num_walls: int = 2, max_num_walls: int = 4, min_num_obstacles: int = 1,
                            max_num_obstacles: int = 2, min_num_goal_objects: int = 1, max_num_goal_objects: int = 1,
                            object_shapes: str = "complex", max_states: int = 2_000_000) -> None:
    """Writes ``puzzle_<i>.pwp`` files (generate.py:136-259, same arguments and checks)."""
    rng = np.random.default_rng(random_seed)
    os.makedirs(save_location_path, exist_ok=True)
    if os.listdir(save_location_path):
        raise ValueError(f"{save_location_path} is not empty!")
    if num_puzzles < 1:
        raise ValueError("num_puzzles must be at least 1")
    if min_puzzle_size < 2 or min_puzzle_size > max_puzzle_size:
        raise ValueError("min_puzzle_size must be >1 and no bigger than max_puzzle_size")
    if min_num_walls < 0 or min_num_walls > max_num_walls:
        raise ValueError("min_num_walls must be >=0 and no bigger than max_num_walls")
    if min_num_obstacles < 0 or min_num_obstacles > max_num_obstacles:
        raise ValueError("min_num_obstacles must be >=0 and no bigger than max_num_obstacles")
    if min_num_goal_objects < 1 or max_num_goal_objects > 2 or min_num_goal_objects > max_num_goal_objects:
        raise ValueError("min_num_goal_objects must be >0, max_num_goal_objects must be <3, and"
                         " min_num_goal_objects must be no bigger than max_num_goal_objects")
    if object_shapes == "simple":
        shapes = SIMPLE_SHAPES
    elif object_shapes == "complex":
        shapes = COMPLEX_SHAPES
    else:
        raise ValueError("object_shapes must be either 'simple' or 'complex'")

    def between(lo, hi):
        return int(rng.integers(lo, hi + 1))

    for i in range(num_puzzles):
        while True:
            try:
                text = generate_puzzle(between(min_puzzle_size, max_puzzle_size), between(min_puzzle_size, max_puzzle_size),
                                       between(min_num_walls, max_num_walls), between(min_num_obstacles, max_num_obstacles),
                                       between(min_num_goal_objects, max_num_goal_objects), shapes, rng)
                break
            except FailedToGenerateError:
                continue
        with open(os.path.join(save_location_path, f"puzzle_{i}{PUZZLE_EXTENSION}"), "w") as f:
            f.write(text)
    if filter_puzzles:
        filter_puzzles_by_solvability(save_location_path, time_limit, num_puzzles, max_states=max_states)
