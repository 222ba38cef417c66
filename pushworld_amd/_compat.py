"""Stand-ins used when ``gym`` / ``dm_env`` are not installed.

The reference pins gym==0.26.1 and dm-env==1.5 (python3/requirements.txt:62,118); neither
contributes arithmetic to the hot path -- only containers (``Discrete``/``Box``,
``TimeStep``/specs).  When the real packages are importable they are used; otherwise these
duck-typed equivalents keep ``PushWorldEnv`` usable (same attribute names and semantics for
the members the reference and its tests touch).
"""
from __future__ import annotations

import enum
from typing import Any, NamedTuple, Optional

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gym as _gym

    HAVE_GYM = True
except Exception:  # noqa: BLE001
    _gym = None
    HAVE_GYM = False

try:  # pragma: no cover
    import dm_env as _dm_env
    from dm_env import specs as _dm_specs

    HAVE_DM_ENV = True
except Exception:  # noqa: BLE001
    _dm_env = None
    _dm_specs = None
    HAVE_DM_ENV = False


# ----------------------------------------------------------------------------- gym
class Discrete:
    def __init__(self, n: int):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64
        self._rng = np.random.default_rng()

    def contains(self, x) -> bool:
        if isinstance(x, (bool, np.bool_)):
            return False
        if isinstance(x, (int, np.integer)):
            v = int(x)
        elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
            v = int(x)
        else:
            return False
        return 0 <= v < self.n

    __contains__ = contains

    def sample(self) -> int:
        return int(self._rng.integers(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high = float(low), float(high)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return bool(
            x.shape == self.shape and np.can_cast(x.dtype, self.dtype) and (x >= self.low).all() and (x <= self.high).all()
        )

    __contains__ = contains

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


if HAVE_GYM:
    GymEnvBase = _gym.Env

    def make_discrete(n):
        return _gym.spaces.Discrete(n)

    def make_box(low, high, shape, dtype):
        return _gym.spaces.Box(low=low, high=high, shape=shape, dtype=dtype)

else:

    class GymEnvBase:  # minimal gym.Env shape
        pass

    def make_discrete(n):
        return Discrete(n)

    def make_box(low, high, shape, dtype):
        return Box(low, high, shape, dtype)


# --------------------------------------------------------------------------- dm_env
class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2


class TimeStep(NamedTuple):
    step_type: Any
    reward: Optional[float]
    discount: Optional[float]
    observation: Any

    def first(self) -> bool:
        return self.step_type == StepType.FIRST

    def mid(self) -> bool:
        return self.step_type == StepType.MID

    def last(self) -> bool:
        return self.step_type == StepType.LAST


class DiscreteArray:
    def __init__(self, num_values, dtype=int, name=None):
        self.num_values = int(num_values)
        self.dtype = np.dtype(dtype)
        self.shape = ()
        self.name = name
        self.minimum = 0
        self.maximum = self.num_values - 1

    def validate(self, value):
        v = np.asarray(value)
        if v.shape != ():
            raise ValueError(f"Expected a scalar action, got shape {v.shape}")
        if not np.issubdtype(v.dtype, np.integer):
            raise ValueError(f"Expected an integer action, got dtype {v.dtype}")
        if not (0 <= int(v) < self.num_values):
            raise ValueError(f"Action {int(v)} is out of bounds [0, {self.num_values})")
        return value


class BoundedArray:
    def __init__(self, shape, dtype, minimum, maximum, name=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.minimum = np.asarray(minimum, dtype=dtype)
        self.maximum = np.asarray(maximum, dtype=dtype)
        self.name = name

    def validate(self, value):
        v = np.asarray(value)
        if v.shape != self.shape or v.dtype != self.dtype:
            raise ValueError("array does not conform to the spec")
        if (v < self.minimum).any() or (v > self.maximum).any():
            raise ValueError("array out of bounds")
        return value


if HAVE_DM_ENV:
    DmEnvBase = _dm_env.Environment
    restart = _dm_env.restart
    transition = _dm_env.transition
    termination = _dm_env.termination
    dm_StepType = _dm_env.StepType

    def make_discrete_array(num_values, dtype, name):
        return _dm_specs.DiscreteArray(num_values=num_values, dtype=dtype, name=name)

    def make_bounded_array(shape, dtype, name, minimum, maximum):
        return _dm_specs.BoundedArray(shape=shape, dtype=dtype, name=name, minimum=minimum, maximum=maximum)

else:

    class DmEnvBase:
        pass

    dm_StepType = StepType

    def restart(observation):
        return TimeStep(StepType.FIRST, None, None, observation)

    def transition(reward, observation, discount=1.0):
        return TimeStep(StepType.MID, reward, discount, observation)

    def termination(reward, observation):
        return TimeStep(StepType.LAST, reward, 0.0, observation)

    def make_discrete_array(num_values, dtype, name):
        return DiscreteArray(num_values, dtype, name)

    def make_bounded_array(shape, dtype, name, minimum, maximum):
        return BoundedArray(shape, dtype, minimum, maximum, name)
