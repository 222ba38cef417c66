"""Vector-environment facades over ``VecPushWorld``.

``PushWorldVectorEnv`` has the call surface of ``gymnasium.vector.VectorEnv`` (``reset`` /
``step`` / ``step_async`` / ``step_wait`` / spaces / ``close``) and ``PushWorldDmVectorEnv`` returns
batched ``dm_env.TimeStep`` tuples.  Per environment the semantics are the reference's
``PushWorldEnv`` (python3/src/pushworld/gym_env.py:57-226, dm_env.py:60-234): same constructor
arguments, rewards, ``terminated`` / ``truncated``; episode turnover follows gymnasium's
next-step autoreset -- the ``step`` after a finished episode ignores the action, draws a new
puzzle on the device (``pw_resample``), resets and returns that puzzle's first observation with
reward 0.  Nothing leaves the GPU unless ``to_numpy=True``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _compat
from .config import PUZZLE_EXTENSION
from .puzzle import DEFAULT_BORDER_WIDTH, DEFAULT_PIXELS_PER_CELL, NUM_ACTIONS, PushWorldPuzzle
from ._single_env import pool_frame
from .utils.filesystem import iter_files_with_extension
from .vec_env import VecPushWorld


class _BatchSpace:
    """Space of ``num_envs`` stacked copies of a single space (``gymnasium.vector.utils.batch_space``)."""

    def __init__(self, single, num_envs: int):
        self.single = single
        self.num_envs = int(num_envs)
        self.shape = (self.num_envs,) + tuple(single.shape)
        self.dtype = single.dtype

    def contains(self, x) -> bool:
        x = np.asarray(x)
        if x.shape != self.shape:
            return False
        if hasattr(self.single, "n"):
            return bool(np.issubdtype(x.dtype, np.integer) and (x >= 0).all() and (x < self.single.n).all())
        return bool((x >= self.single.low).all() and (x <= self.single.high).all())

    __contains__ = contains

    def sample(self):
        if hasattr(self.single, "n"):
            return np.random.default_rng().integers(self.single.n, size=self.shape)
        raise NotImplementedError

    def __repr__(self):
        return f"Batch({self.single!r}, {self.num_envs})"


def _load_pool(puzzle_path, standard_padding: bool):
    """The puzzle pool and padded frame of ``PushWorldEnv.__init__`` (gym_env.py:66-103)."""
    if isinstance(puzzle_path, (list, tuple)):
        puzzles = [p if isinstance(p, PushWorldPuzzle) else PushWorldPuzzle(p) for p in puzzle_path]
    else:
        puzzles = [PushWorldPuzzle(p) for p in iter_files_with_extension(puzzle_path, PUZZLE_EXTENSION)]
    if len(puzzles) == 0:
        raise ValueError(f"No PushWorld puzzles found in: {puzzle_path}")
    return puzzles, pool_frame(puzzles, standard_padding)


class _VectorCore:
    def __init__(self, puzzle_path, num_envs: int, max_steps: Optional[int], border_width: int, pixels_per_cell: int,
                 standard_padding: bool, observation: str, seed: int, sample_table, device: Optional[int],
                 to_numpy: bool, incremental: bool = True):
        if border_width < 1:
            raise ValueError("border_width must be >= 1")
        if pixels_per_cell < 3:
            raise ValueError("pixels_per_cell must be >= 3")
        if observation not in ("float32", "uint8"):
            raise ValueError("observation must be 'float32' or 'uint8'")
        puzzles, pad = _load_pool(puzzle_path, standard_padding)
        self.vec = VecPushWorld(puzzles, num_envs, max_steps=max_steps, border_width=border_width,
                                pixels_per_cell=pixels_per_cell, observation=observation, pad_cells=pad,
                                device=device, autoreset=True, resample=True if sample_table is None else sample_table,
                                seed=seed, incremental=incremental)
        self.num_envs = int(num_envs)
        self.to_numpy = bool(to_numpy)
        self._obs_dtype = np.float32 if observation == "float32" else np.uint8
        self._obs_high = 1.0 if observation == "float32" else 255
        self._pending = False
        self._closed = False

    @property
    def puzzles(self):
        return self.vec.puzzles

    @property
    def puzzle_ids(self) -> torch.Tensor:
        """int32 [num_envs]: index (into ``puzzles``) of each environment's current puzzle."""
        return self.vec.puzzle_id

    def _actions(self, actions) -> torch.Tensor:
        if isinstance(actions, torch.Tensor):
            a = actions
            if a.shape != (self.num_envs,) or a.dtype in (torch.bool, torch.float16, torch.float32, torch.float64, torch.bfloat16):
                raise ValueError("The provided action is not in the action space.")
            if a.device != self.vec.device or a.dtype != torch.uint8:
                # Wider integer types are range-checked BEFORE the cast (256 would become LEFT, -1 would become
                # 255): one fused reduction where the tensor lives; on the device this synchronises, which the
                # conversion path can afford -- pass uint8 device tensors to stay asynchronous.
                if bool(((a < 0) | (a >= NUM_ACTIONS)).any()):
                    raise ValueError("The provided action is not in the action space.")
                a = a.to(device=self.vec.device, dtype=torch.uint8)
            return a
        arr = np.asarray(actions)
        if arr.shape != (self.num_envs,) or not np.issubdtype(arr.dtype, np.integer) or (arr < 0).any() or (arr >= NUM_ACTIONS).any():
            raise ValueError("The provided action is not in the action space.")
        return torch.as_tensor(arr.astype(np.uint8)).to(self.vec.device)

    def _out(self, t: torch.Tensor):
        return t.cpu().numpy() if self.to_numpy else t

    def _launch(self, actions) -> None:
        if self._pending:
            raise RuntimeError("step_async() was called twice without step_wait()")
        self.vec.step(self._actions(actions))  # asynchronous on the current HIP stream
        self._pending = True

    def _collect(self):
        if not self._pending:
            raise RuntimeError("step_wait() called without a pending step_async()")
        self._pending = False
        v = self.vec
        if self.to_numpy:  # the .cpu() copies synchronise; device tensors stay stream-ordered
            bad = v.terminated.cpu().numpy() == 0xFF
            if bad.any():  # device-side action check of pw_step (action outside 0..3)
                v.engine.bad_actions()  # reported here: clear the sticky counter
                raise ValueError("The provided action is not in the action space.")
        return v.obs, v.reward, v.terminated, v.truncated

    def check_actions(self) -> None:
        """Device-tensor mode (``to_numpy=False``) never synchronises, so a uint8 action outside 0..3 cannot raise
        inside ``step``: the kernel leaves that environment untouched, flags it 0xFF (the next step then resets
        it) and counts it in the engine's sticky counter.  This reads and clears the counter (one stream
        synchronisation) and raises the reference's ``ValueError`` (gym_env.py:195-196) if any was seen."""
        n = self.vec.engine.bad_actions()
        if n:
            raise ValueError(f"The provided action is not in the action space. ({n} since the last check)")

    def close(self) -> None:
        self._closed = True


class PushWorldVectorEnv(_VectorCore):
    """``gymnasium.vector.VectorEnv``-style batch of ``PushWorldEnv``s on one MI355X.

    Args (first five as ``PushWorldEnv``, gym_env.py:57-64):
        puzzle_path: ``.pwp`` file, directory searched recursively, or a list of paths / puzzles.
        num_envs: batch size B.
        observation: "float32" (reference observation, values in [0, 1]) or "uint8" (4x fewer bytes).
        seed: seed of the per-episode puzzle draw.
        sample_table: optional pool indices to draw from (repeats = weights); default uniform.
        to_numpy: return host numpy arrays (and python-side checks) instead of device tensors.
    """

    metadata = {"render_modes": ["rgb_array"], "autoreset_mode": "next_step"}
    render_mode = "rgb_array"

    def __init__(self, puzzle_path, num_envs: int, max_steps: Optional[int] = None,
                 border_width: int = DEFAULT_BORDER_WIDTH, pixels_per_cell: int = DEFAULT_PIXELS_PER_CELL,
                 standard_padding: bool = False, observation: str = "float32", seed: int = 0, sample_table=None,
                 device: Optional[int] = None, to_numpy: bool = False):
        super().__init__(puzzle_path, num_envs, max_steps, border_width, pixels_per_cell, standard_padding,
                         observation, seed, sample_table, device, to_numpy)
        self.single_action_space = _compat.make_discrete(NUM_ACTIONS)
        self.single_observation_space = _compat.make_box(0, self._obs_high, self.vec.engine.obs_shape, self._obs_dtype)
        self.action_space = _BatchSpace(_compat.Discrete(NUM_ACTIONS), num_envs)
        self.observation_space = _BatchSpace(_compat.Box(0, self._obs_high, self.vec.engine.obs_shape, self._obs_dtype),
                                             num_envs)

    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        """All environments draw a puzzle and start an episode (gym_env.py:150-186).  Returns
        ``(obs [B, H, W, 3], {"puzzle_id": int32 [B]})``."""
        self._pending = False
        obs = self.vec.reset(seed=seed)
        return self._out(obs), self._info()

    def _info(self) -> dict:
        """``puzzle_state``: int8 [B, NP, 2] (x, y) positions, agent first, zero beyond a puzzle's movables
        (the batched form of the reference's ``info["puzzle_state"]``, gym_env.py:186,226)."""
        return {"puzzle_id": self._out(self.vec.puzzle_id), "puzzle_state": self._out(self.vec.pos)}

    def step_async(self, actions) -> None:
        self._launch(actions)

    def step_wait(self):
        obs, reward, terminated, truncated = self._collect()
        info = self._info()
        if self.to_numpy:
            return (obs.cpu().numpy(), reward.cpu().numpy(), terminated.cpu().numpy().astype(bool),
                    truncated.cpu().numpy().astype(bool), info)
        return obs, reward, terminated, truncated, info

    def step(self, actions):
        """gym_env.py:188-226 for every environment -> ``(obs, reward f64[B], terminated, truncated, info)``."""
        self.step_async(actions)
        return self.step_wait()

    def render(self):
        """uint8 frames [B, H, W, 3] of the current states (padded frame)."""
        if self.vec.observation == "uint8":
            return self._out(self.vec.obs)
        return self._out((self.vec.obs * 255.0).round().to(torch.uint8))


class PushWorldDmVectorEnv(_VectorCore):
    """Batched ``dm_env`` flavour: ``reset`` / ``step`` return one ``TimeStep`` whose fields are
    [B]-shaped (``step_type`` int32, ``reward`` float64, ``discount`` float64) plus the observation
    batch.  Per environment: FIRST after a (auto)reset with reward 0 / discount 1 (``dm_env.restart``
    carries None there; arrays cannot), LAST with discount 0 when terminated or truncated
    (dm_env.py:229-232), MID otherwise."""

    def __init__(self, puzzle_path, num_envs: int, max_steps: Optional[int] = None,
                 border_width: int = DEFAULT_BORDER_WIDTH, pixels_per_cell: int = DEFAULT_PIXELS_PER_CELL,
                 standard_padding: bool = False, observation: str = "float32", seed: int = 0, sample_table=None,
                 device: Optional[int] = None, to_numpy: bool = False):
        super().__init__(puzzle_path, num_envs, max_steps, border_width, pixels_per_cell, standard_padding,
                         observation, seed, sample_table, device, to_numpy)
        self._action_spec = _compat.make_discrete_array(NUM_ACTIONS, int, "action")
        self._observation_spec = _compat.make_bounded_array(self.vec.engine.obs_shape, self._obs_dtype, "board", 0,
                                                            self._obs_high)
        self._was_done = None

    def action_spec(self):
        return self._action_spec

    def observation_spec(self):
        return self._observation_spec

    def _timestep(self, step_type, reward, discount, obs):
        return _compat.TimeStep(self._out(step_type), self._out(reward), self._out(discount), self._out(obs))

    def reset(self, seed: Optional[int] = None):
        self._pending = False
        obs = self.vec.reset(seed=seed)
        dev = self.vec.device
        self._was_done = torch.zeros((self.num_envs,), dtype=torch.bool, device=dev)
        return self._timestep(torch.full((self.num_envs,), int(_compat.StepType.FIRST), dtype=torch.int32, device=dev),
                              torch.zeros((self.num_envs,), dtype=torch.float64, device=dev),
                              torch.ones((self.num_envs,), dtype=torch.float64, device=dev), obs)

    def step(self, actions):
        if self._was_done is None:
            raise RuntimeError("reset() must be called before step() can be called.")
        self._launch(actions)
        obs, reward, terminated, truncated = self._collect()
        done = (terminated != 0) | (truncated != 0)
        first = self._was_done  # this call reset those environments
        step_type = torch.where(first, int(_compat.StepType.FIRST),
                                torch.where(done, int(_compat.StepType.LAST), int(_compat.StepType.MID))).to(torch.int32)
        discount = torch.where(done & ~first, 0.0, 1.0).to(torch.float64)
        self._was_done = done & ~first
        return self._timestep(step_type, reward, discount, obs)
