"""The 8 dihedral variants of a puzzle (same surface as python3/src/pushworld/transform.py).

Used to widen a training pool: the variants are ordinary puzzles, so they go through the normal
host compiler into the packed set (8 x the tables, still a few MB per thousand puzzles).  Names and
output formatting follow the reference: ``r{0,90,180,270}`` = clockwise rotation, ``_flipped`` =
top-bottom flip applied BEFORE the rotation; tokens joined by two spaces, rows by a newline.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

from .config import PUZZLE_EXTENSION

TRANSFORM_NAMES = tuple(f"r{rot}{suffix}" for suffix in ("", "_flipped") for rot in (0, 90, 180, 270))

# action ids: 0 LEFT, 1 RIGHT, 2 UP, 3 DOWN (puzzle.py:43-56)
_ROT90 = {2: 1, 1: 3, 3: 0, 0: 2}   # one clockwise quarter turn: UP -> RIGHT -> DOWN -> LEFT -> UP
_FLIP = {2: 3, 3: 2, 0: 0, 1: 1}    # top-bottom flip


def _rotate_clockwise(grid: List[List[str]]) -> List[List[str]]:
    rows, cols = len(grid), len(grid[0])
    return [[grid[rows - 1 - c][r] for c in range(rows)] for r in range(cols)]


def get_puzzle_transforms(puzzle_string: str) -> Dict[str, str]:
    """All 8 combinations of quarter turns and a flip of a ``.pwp`` puzzle text (transform.py:21-48)."""
    base = [line.split() for line in puzzle_string.splitlines()]
    out = {}
    for suffix, grid in (("", base), ("_flipped", base[::-1])):
        for rot in (0, 90, 180, 270):
            out[f"r{rot}{suffix}"] = "\n".join("  ".join(row) for row in grid)
            grid = _rotate_clockwise(grid)
    return out


def transform_grids(grids, dims):
    """The 8 variants of every symbol grid ON THE DEVICE (``pw_transform_grids``): ``grids`` uint8 [n, S * S] and
    ``dims`` int32 [n, 2] device tensors -> ``(uint8 [8 n, S * S], int32 [8 n, 2])``, variant ``v`` of puzzle ``i``
    at row ``8 i + v`` in the order of ``TRANSFORM_NAMES``."""
    import ctypes

    import torch

    from . import _capi

    n, cells = grids.shape
    slot_w = int(round(cells ** 0.5))
    if slot_w * slot_w != cells or dims.shape != (n, 2) or grids.dtype != torch.uint8 or dims.dtype != torch.int32:
        raise ValueError("grids must be uint8 [n, S * S] and dims int32 [n, 2]")
    out = torch.empty((n * 8, cells), dtype=torch.uint8, device=grids.device)
    out_dims = torch.empty((n * 8, 2), dtype=torch.int32, device=grids.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(grids.device).cuda_stream)
    _capi.check(_capi.lib.pw_transform_grids(grids.device.index, _capi._ptr(grids), _capi._ptr(dims), n, slot_w,
                                             _capi._ptr(out), _capi._ptr(out_dims), stream))
    return out, out_dims


def transform_plan(plan: Sequence[int], transform_name: str) -> List[int]:
    """The plan that does in the transformed puzzle what ``plan`` does in the original
    (the mapping python3/test/test_transform.py:47-79 checks)."""
    if transform_name not in TRANSFORM_NAMES:
        raise ValueError(f"unknown transform {transform_name!r}")
    actions = [int(a) for a in plan]
    if transform_name.endswith("_flipped"):
        actions = [_FLIP[a] for a in actions]
    for _ in range(int(transform_name[1:].split("_")[0]) // 90):
        actions = [_ROT90[a] for a in actions]
    return actions


def create_transformed_puzzles(puzzle_path: str, output_path: str) -> None:
    """Writes the 8 variants of every puzzle under ``puzzle_path`` to ``output_path`` as
    ``<relative name>_<transform>.pwp`` (transform.py:51-85)."""
    root = puzzle_path.rstrip(os.path.sep)
    for subdir, _, filenames in os.walk(root):
        for filename in filenames:
            if not filename.endswith(PUZZLE_EXTENSION):
                continue
            src = os.path.join(subdir, filename)
            with open(src, "r") as f:
                variants = get_puzzle_transforms(f.read())
            prefix = os.path.splitext(src[len(root) + 1:])[0]
            for name, text in variants.items():
                dst = os.path.join(output_path, f"{prefix}_{name}{PUZZLE_EXTENSION}")
                os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
                with open(dst, "w") as f:
                    f.write(text)
