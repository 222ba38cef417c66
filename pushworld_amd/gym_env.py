"""Gym adapter with the reference surface (python3/src/pushworld/gym_env.py:29-240)."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np

from ._compat import GymEnvBase, make_box, make_discrete
from ._single_env import SingleEnvCore
from .puzzle import DEFAULT_BORDER_WIDTH, DEFAULT_PIXELS_PER_CELL, NUM_ACTIONS, PushWorldPuzzle


class PushWorldEnv(SingleEnvCore, GymEnvBase):
    """An OpenAI-Gym style environment for PushWorld puzzles, stepped and rendered on an MI355X.

    Same constructor arguments, return values, reward shaping and error behaviour as the
    reference ``pushworld.gym_env.PushWorldEnv``.
    """

    def __init__(self, puzzle_path: str, max_steps: Optional[int] = None,
                 border_width: int = DEFAULT_BORDER_WIDTH, pixels_per_cell: int = DEFAULT_PIXELS_PER_CELL,
                 standard_padding: bool = False) -> None:
        SingleEnvCore.__init__(self, puzzle_path, max_steps, border_width, pixels_per_cell, standard_padding)
        self._action_space = make_discrete(NUM_ACTIONS)
        self._observation_space = make_box(0.0, 1.0, self.obs_shape, np.float32)

    @property
    def action_space(self):
        return self._action_space

    @property
    def observation_space(self):
        return self._observation_space

    @property
    def metadata(self) -> Dict[str, Any]:
        return {"render_modes": ["rgb_array"]}

    @property
    def render_mode(self) -> str:
        return "rgb_array"

    @property
    def current_puzzle(self) -> Optional[PushWorldPuzzle]:
        return self._current_puzzle

    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None) -> Tuple[np.ndarray, dict]:
        """gym_env.py:150-186."""
        observation = self.core_reset(seed)
        return observation, {"puzzle_state": self._current_state}

    def step(self, action: int):
        """gym_env.py:188-226."""
        if not self._action_space.contains(action):
            raise ValueError("The provided action is not in the action space.")
        observation, reward, terminated, truncated = self.core_step(action)
        return observation, reward, terminated, truncated, {"puzzle_state": self._current_state}

    def render(self, mode="rgb_array") -> np.ndarray:
        """gym_env.py:228-240: uint8, unpadded."""
        assert mode == "rgb_array", "mode must be rgb_array."
        return self.core_render_u8()
