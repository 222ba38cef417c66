"""dm_env adapter with the reference surface (python3/src/pushworld/dm_env.py:35-252)."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _compat
from ._single_env import SingleEnvCore
from .puzzle import DEFAULT_BORDER_WIDTH, DEFAULT_PIXELS_PER_CELL, NUM_ACTIONS, PushWorldPuzzle, State


class PushWorldEnv(SingleEnvCore, _compat.DmEnvBase):
    """A dm_env style environment for PushWorld puzzles, stepped and rendered on an MI355X."""

    def __init__(self, puzzle_path: str, max_steps: Optional[int] = None,
                 border_width: int = DEFAULT_BORDER_WIDTH, pixels_per_cell: int = DEFAULT_PIXELS_PER_CELL,
                 standard_padding: bool = False) -> None:
        SingleEnvCore.__init__(self, puzzle_path, max_steps, border_width, pixels_per_cell, standard_padding)
        self._action_space = _compat.make_discrete_array(NUM_ACTIONS, int, "action")
        self._observation_space = _compat.make_bounded_array(self.obs_shape, np.float32, "board", 0.0, 1.0)

    def observation_spec(self):
        return self._observation_space

    def action_spec(self):
        return self._action_space

    @property
    def current_puzzle(self) -> Optional[PushWorldPuzzle]:
        return self._current_puzzle

    @property
    def current_state(self) -> Optional[State]:
        return self._current_state

    def reset(self, seed: Optional[int] = None):
        """dm_env.py:150-183 -> ``restart(observation)``."""
        return _compat.restart(self.core_reset(seed))

    def step(self, action: int):
        """dm_env.py:185-234: LAST (discount 0) for terminated or truncated."""
        try:
            self._action_space.validate(action)
        except ValueError:
            raise ValueError("The provided action is not in the action space.")
        observation, reward, terminated, truncated = self.core_step(action)
        if terminated or truncated:
            return _compat.termination(reward, observation)
        return _compat.transition(reward, observation)

    def render(self, mode="rgb_array") -> np.ndarray:
        """dm_env.py:236-251: float32 / 255, unpadded."""
        assert mode == "rgb_array", "mode must be rgb_array."
        return self.core_render_u8().astype(np.float32) / 255
