"""Observation helpers (reference: python3/src/pushworld/utils/env_utils.py:25-91)."""
from typing import Tuple

import numpy as np

from pushworld_amd import _capi
from pushworld_amd.config import BENCHMARK_PUZZLES_PATH, PUZZLE_EXTENSION
from pushworld_amd.utils.filesystem import iter_files_with_extension


def get_max_puzzle_dimensions() -> Tuple[int, int]:
    """(max height, max width) in cells over the benchmark puzzle files, env_utils.py:25-41.
    Like the reference it only sees extracted ``.pwp`` files (levels 1-4), not the level-0 zip."""
    max_height = 0
    max_width = 0
    for path in iter_files_with_extension(BENCHMARK_PUZZLES_PATH, PUZZLE_EXTENSION):
        with open(path, "r") as f:
            lines = f.readlines()
        max_height = max(max_height, len(lines) + 2)
        max_width = max(max_width, len(lines[0].strip().split()) + 2)
    return max_height, max_width


def render_observation_padded(puzzle, state, max_cell_height: int, max_cell_width: int, pixels_per_cell: int,
                              border_width: int) -> np.ndarray:
    """env_utils.py:44-91 on the GPU: float32 image in [0, 1], zero padded and centred."""
    if border_width < 1:
        raise ValueError("border_width must be >= 1")
    if pixels_per_cell < 1 + 2 * border_width:
        raise ValueError("pixels_per_cell must be >= 1 + 2*border_width")
    key = ("padded", max_cell_height, max_cell_width, pixels_per_cell, border_width)
    eng = puzzle._engines.get(key)
    if eng is None:
        eng = _capi.Engine(puzzle._puzzle_set(), None, pixels_per_cell, border_width, _capi.OBS_F32,
                           max_cell_height, max_cell_width)
        puzzle._engines[key] = eng
    b = puzzle._state_bufs(eng)
    puzzle._upload(eng, state)
    if key not in b:
        b[key] = eng.alloc_obs(1)
    storage, view = b[key]
    eng.render(b["pid"], b["pos"], storage)
    return view[0].cpu().numpy()
