"""Single-environment core shared by the gym and dm_env adapters.

Reproduces the constructor / reset / step bookkeeping of
python3/src/pushworld/gym_env.py:57-226 and dm_env.py:60-234 around a batch-1 engine.  The
puzzle pool is ONE packed puzzle set in HBM; ``reset`` only changes the environment's
puzzle id.
"""
from __future__ import annotations

import random
from typing import Optional

import numpy as np
import torch

from . import _capi
from .config import PUZZLE_EXTENSION
from .puzzle import NUM_ACTIONS, PushWorldPuzzle, default_device_index
from .utils.env_utils import get_max_puzzle_dimensions
from .utils.filesystem import iter_files_with_extension


class SingleEnvCore:
    def __init__(self, puzzle_path: str, max_steps: Optional[int], border_width: int, pixels_per_cell: int,
                 standard_padding: bool) -> None:
        self._puzzles = [PushWorldPuzzle(p) for p in iter_files_with_extension(puzzle_path, PUZZLE_EXTENSION)]
        if len(self._puzzles) == 0:
            raise ValueError(f"No PushWorld puzzles found in: {puzzle_path}")
        if border_width < 1:
            raise ValueError("border_width must be >= 1")
        if pixels_per_cell < 3:
            raise ValueError("pixels_per_cell must be >= 3")
        self._max_steps = max_steps
        self._pixels_per_cell = pixels_per_cell
        self._border_width = border_width

        widths, heights = zip(*[p.dimensions for p in self._puzzles])
        self._max_cell_width = max(widths)
        self._max_cell_height = max(heights)
        if standard_padding:
            std_h, std_w = get_max_puzzle_dimensions()
            if std_h < self._max_cell_height:
                raise ValueError(
                    "`standard_padding` is True, but the maximum puzzle height in BENCHMARK_PUZZLES_PATH is "
                    "less than the height of the puzzle(s) in the given `puzzle_path`."
                )
            self._max_cell_height = std_h
            if std_w < self._max_cell_width:
                raise ValueError(
                    "`standard_padding` is True, but the maximum puzzle width in BENCHMARK_PUZZLES_PATH is "
                    "less than the width of the puzzle(s) in the given `puzzle_path`."
                )
            self._max_cell_width = std_w

        # gym_env.py:107-109: fixed seed for reproducibility
        self._random_generator = random.Random(123)
        self._current_puzzle = None
        self._current_state = None
        self._steps = 0

        dev = default_device_index()
        self._pset = _capi.PuzzleSet([p._parsed for p in self._puzzles], dev)
        # raises ValueError("pixels_per_cell must be >= 1 + 2*border_width") like the render
        # call inside the reference constructor (gym_env.py:116-123 -> puzzle.py:447-448)
        self._engine = _capi.Engine(self._pset, max_steps, pixels_per_cell, border_width, _capi.OBS_F32,
                                    self._max_cell_height, self._max_cell_width)
        self._render_engines = {}
        st = self._engine.alloc_state(1)
        self._buf = st
        self._pid = torch.zeros((1,), dtype=torch.int32, device=self._engine.device)
        self._act = torch.zeros((1,), dtype=torch.uint8, device=self._engine.device)
        self._obs_storage, self._obs = self._engine.alloc_obs(1)
        self.obs_shape = self._engine.obs_shape

    # ------------------------------------------------------------------
    def _state_from_device(self):
        n = self._current_puzzle.num_movables
        arr = self._buf["pos"][0, :n].cpu().tolist()
        return tuple((int(x), int(y)) for x, y in arr)

    def core_reset(self, seed: Optional[int]) -> np.ndarray:
        if seed is not None:
            self._random_generator = random.Random(seed)
        self._current_puzzle = self._random_generator.choice(self._puzzles)
        index = next(i for i, p in enumerate(self._puzzles) if p is self._current_puzzle)
        self._pid.fill_(index)
        b = self._buf
        self._engine.reset(self._pid, b["pos"], b["steps"], b["terminated"], b["truncated"])
        self._engine.render(self._pid, b["pos"], self._obs_storage)
        self._current_state = self._current_puzzle.initial_state
        self._current_achieved_goals = self._current_puzzle.count_achieved_goals(self._current_state)
        self._steps = 0
        return self._obs[0].cpu().numpy()

    def core_step(self, action: int):
        """Returns (observation, reward: float, terminated: bool, truncated: bool)."""
        if self._current_state is None:
            raise RuntimeError("reset() must be called before step() can be called.")
        b = self._buf
        self._act.fill_(int(action))
        self._engine.step_render(self._pid, self._act, b["pos"], b["steps"], b["reward"], b["dgoals"],
                                 b["terminated"], b["truncated"], self._obs_storage)
        observation = self._obs[0].cpu().numpy()  # synchronises the stream
        self._steps += 1
        self._current_state = self._state_from_device()
        reward = float(b["reward"].cpu()[0])
        terminated = bool(b["terminated"].cpu()[0])
        truncated = bool(b["truncated"].cpu()[0])
        return observation, reward, terminated, truncated

    def core_render_u8(self) -> np.ndarray:
        """uint8, unpadded: puzzle.render(current_state) (gym_env.py:228-240)."""
        return self._current_puzzle.render(self._current_state, border_width=self._border_width,
                                           pixels_per_cell=self._pixels_per_cell)


def validate_num_actions():
    return NUM_ACTIONS
