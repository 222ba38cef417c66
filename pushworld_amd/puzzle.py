"""``PushWorldPuzzle`` with the reference's Python surface, executed on an MI355X.

Mirrors ``python3/src/pushworld/puzzle.py`` of google-deepmind/pushworld (class surface
:100-128, properties :313-346, methods :348-506) so user code and the reference's own tests
can switch imports.  Parsing happens in the C++ host library, dynamics and rendering in the
HIP kernels of ``libpushworld_amd.so``; there is no CPU implementation of either -- calling
``get_next_state`` / ``render`` without a visible HIP device raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from itertools import chain
from typing import Iterable, List, Optional, Set, Tuple

import numpy as np
import torch

from . import _capi

# puzzle.py:22-29
DEFAULT_BORDER_WIDTH = 2
DEFAULT_PIXELS_PER_CELL = 20
NUM_ACTIONS = 4
AGENT_IDX = 0

Point = Tuple[int, int]
State = Tuple[Point, ...]
Color = Tuple[int, int, int]


class Actions:
    """puzzle.py:32-50."""

    LEFT, RIGHT, UP, DOWN = range(NUM_ACTIONS)
    FROM_CHAR = {"L": LEFT, "R": RIGHT, "U": UP, "D": DOWN}
    DISPLACEMENTS = np.array([(-1, 0), (1, 0), (0, -1), (0, 1)])


def hex_to_rgb(hex_string: str) -> Color:
    return tuple(int(hex_string[i : i + 2], 16) for i in (0, 2, 4))


class Colors:
    """puzzle.py:65-79 (the kernels carry the same palette, csrc/pw_kernels.hip)."""

    AGENT = hex_to_rgb("00DC00")
    AGENT_BORDER = hex_to_rgb("006E00")
    AGENT_WALL = hex_to_rgb("FAC71E")
    AGENT_WALL_BORDER = hex_to_rgb("7D640F")
    GOAL = None
    GOAL_BORDER = hex_to_rgb("B90000")
    GOAL_OBJECT = hex_to_rgb("DC0000")
    GOAL_OBJECT_BORDER = hex_to_rgb("6E0000")
    MOVABLE = hex_to_rgb("469BFF")
    MOVABLE_BORDER = hex_to_rgb("23487F")
    WALL = hex_to_rgb("0A0A0A")
    WALL_BORDER = hex_to_rgb("050505")


@dataclass(frozen=True)
class PushWorldObject:
    """puzzle.py:82-97."""

    position: Point
    fill_color: Optional[Color]
    border_color: Color
    cells: Set[Point]


def default_device_index() -> int:
    """HIP device used by the single-environment wrappers (``PUSHWORLD_AMD_DEVICE``)."""
    if "PUSHWORLD_AMD_DEVICE" in os.environ:
        return int(os.environ["PUSHWORLD_AMD_DEVICE"])
    if not torch.cuda.is_available() or _capi.device_count() == 0:
        raise RuntimeError(
            "pushworld_amd needs a HIP device (MI355X / gfx950); none is visible and there is no CPU fallback"
        )
    return torch.cuda.current_device()


def _as_state(state) -> State:
    return tuple((int(p[0]), int(p[1])) for p in state)


class PushWorldPuzzle:
    """A puzzle in the PushWorld environment (reference surface, GPU execution).

    Args:
        file_path: path of a ``.pwp`` file.
        text: alternatively the file contents (keyword only).
        order: ``"python"`` (default, puzzle.py object order) or ``"cpp"``
            (pushworld_puzzle.cc object order), see SURVEY trap T1.
    """

    def __init__(self, file_path: Optional[str] = None, *, text: Optional[str] = None, order: str = "python"):
        if text is None:
            with open(file_path, "r") as fi:
                text = fi.read()
        self.file_path = file_path
        self._text = text
        self._parsed = _capi.ParsedPuzzle(text, _capi.ORDER_PYTHON if order == "python" else _capi.ORDER_CPP)
        p = self._parsed
        self._width, self._height = p.width, p.height
        self.num_movables = p.num_movables
        self._initial_state: State = p.initial_state
        self._goal_state = p.goal_state
        self._lazy_props = {}
        self._pset = None
        self._engines = {}
        self._bufs = None
        self._batch = None  # state buffers of get_next_states
        self._lat = None    # pre-marshalled pw_next_state call

    # ---------------------------------------------------------------- properties
    @property
    def initial_state(self) -> State:
        return self._initial_state

    @property
    def goal_state(self) -> Tuple[Point, ...]:
        return self._goal_state

    @property
    def dimensions(self) -> Tuple[int, int]:
        return (self._width, self._height)

    def _prop(self, key, make):
        if key not in self._lazy_props:
            self._lazy_props[key] = make()
        return self._lazy_props[key]

    @property
    def wall_positions(self) -> Set[Point]:
        return self._prop("wall", lambda: set(self._parsed.wall_cells))

    @property
    def agent_wall_positions(self) -> Set[Point]:
        # trap T2: the reference property returns AW u W (puzzle.py:253,273,338-341)
        return self._prop("aw", lambda: set(self._parsed.agent_wall_cells) | self.wall_positions)

    @property
    def movable_objects(self) -> List[PushWorldObject]:
        def make():
            p, g, out = self._parsed, self._parsed.num_goals, []
            for j in range(p.num_movables):
                if j == 0:
                    fill, border = Colors.AGENT, Colors.AGENT_BORDER
                elif j <= g:
                    fill, border = Colors.GOAL_OBJECT, Colors.GOAL_OBJECT_BORDER
                else:
                    fill, border = Colors.MOVABLE, Colors.MOVABLE_BORDER
                out.append(PushWorldObject(position=p.initial_state[j], fill_color=fill, border_color=border,
                                           cells=set(p.object_cells[j])))
            return out

        return self._prop("movables", make)

    @property
    def _goals(self) -> List[PushWorldObject]:
        p = self._parsed
        return self._prop("goals", lambda: [
            PushWorldObject(position=p.goal_state[k], fill_color=Colors.GOAL, border_color=Colors.GOAL_BORDER,
                            cells=set(p.goal_cells[k])) for k in range(p.num_goals)])

    # ---------------------------------------------------------------- device plumbing
    def _puzzle_set(self) -> "_capi.PuzzleSet":
        if self._pset is None:
            self._pset = _capi.PuzzleSet([self._parsed], default_device_index())
        return self._pset

    def _engine(self, ppc=3, bw=1, dtype=_capi.OBS_U8) -> "_capi.Engine":
        key = (ppc, bw, dtype)
        eng = self._engines.get(key)
        if eng is None:
            eng = _capi.Engine(self._puzzle_set(), None, ppc, bw, dtype)
            self._engines[key] = eng
        return eng

    def _state_bufs(self, eng):
        if self._bufs is None:
            self._bufs = eng.alloc_state(1)
            self._bufs["pid"] = torch.zeros((1,), dtype=torch.int32, device=eng.device)
            self._bufs["act"] = torch.zeros((1,), dtype=torch.uint8, device=eng.device)
            # staging of the render path: pinned host copy of the position row, uploaded with one asynchronous copy
            self._bufs["pos_host"] = torch.zeros((1, eng.np, 2), dtype=torch.int8).pin_memory()
        return self._bufs

    def _state_bytes(self, state) -> bytes:
        """The 2 N int8 values of a state as bytes (the wire format of ``pw_next_state``)."""
        if len(state) != self.num_movables:
            raise ValueError(f"state has {len(state)} positions, puzzle has {self.num_movables} movables")
        try:
            flat = bytes(chain.from_iterable(state))
        except (ValueError, TypeError):  # negative or non-integer coordinates: int8 two's complement like the kernels read them
            flat = bytes(int(v) & 0xFF for v in chain.from_iterable(state))
        if len(flat) != 2 * self.num_movables:
            raise ValueError("every position of a state is an (x, y) pair")
        return flat

    def _upload(self, eng, state) -> None:
        b = self._state_bufs(eng)
        flat = self._state_bytes(state)
        host = b["pos_host"]
        host.view(torch.uint8).view(-1)[: len(flat)] = torch.frombuffer(bytearray(flat), dtype=torch.uint8)
        b["pos"].copy_(host, non_blocking=True)

    # ---------------------------------------------------------------- dynamics
    def get_next_state(self, state: State, action: int) -> State:
        """puzzle.py:348-394 on the GPU: ``pw_next_state`` -- the state travels in the kernel arguments, one wavefront
        computes the step, the result lands in pinned host memory and the call returns when the kernel's completion
        word arrives (one launch, no copy command, no stream synchronisation)."""
        if action not in (0, 1, 2, 3):
            raise ValueError("action must be one of 0 (L), 1 (R), 2 (U), 3 (D)")
        lat = self._lat
        if lat is None:
            eng = self._engine()
            out = ctypes.create_string_buffer(64)
            lat = self._lat = (_capi.lib.pw_next_state, eng.handle, out, memoryview(out).cast("b"), 2 * self.num_movables, eng)
        fn, handle, out, view, n2, _ = lat
        rc = fn(handle, 0, self._state_bytes(state), int(action), out, None)
        if rc:
            _capi.check(rc)
        return tuple(zip(view[0:n2:2], view[1:n2:2]))

    def get_next_states(self, states, actions) -> np.ndarray:
        """The batched sibling of ``get_next_state``: ``states`` int array [B, N, 2], ``actions`` int array [B] ->
        next states int8 [B, N, 2].  One asynchronous upload from pinned memory, one ``pw_step`` launch, one download."""
        st = np.ascontiguousarray(np.asarray(states), dtype=np.int8)
        acts = np.ascontiguousarray(np.asarray(actions), dtype=np.uint8)
        if st.ndim != 3 or st.shape[1:] != (self.num_movables, 2) or acts.shape != (st.shape[0],):
            raise ValueError(f"states must have shape [B, {self.num_movables}, 2] and actions shape [B]")
        if acts.size and acts.max() > 3:
            raise ValueError("action must be one of 0 (L), 1 (R), 2 (U), 3 (D)")
        B = st.shape[0]
        if B == 0:
            return np.zeros((0, self.num_movables, 2), np.int8)
        eng = self._engine()
        if self._batch is None or self._batch["pos"].shape[0] < B:
            cap = max(B, 1024)
            bufs = eng.alloc_state(cap)
            bufs["pid"] = torch.zeros((cap,), dtype=torch.int32, device=eng.device)
            bufs["act"] = torch.zeros((cap,), dtype=torch.uint8, device=eng.device)
            bufs["pos_host"] = torch.zeros((cap, eng.np, 2), dtype=torch.int8).pin_memory()
            bufs["act_host"] = torch.zeros((cap,), dtype=torch.uint8).pin_memory()
            self._batch = bufs
        b = self._batch
        ph, ah = b["pos_host"], b["act_host"]
        ph[:B].zero_()
        ph[:B, : self.num_movables] = torch.from_numpy(st)
        ah[:B] = torch.from_numpy(acts)
        pos, act = b["pos"][:B], b["act"][:B]
        pos.copy_(ph[:B], non_blocking=True)
        act.copy_(ah[:B], non_blocking=True)
        eng.step(b["pid"][:B], act, pos, b["steps"][:B], b["reward"][:B], b["dgoals"][:B], b["terminated"][:B], b["truncated"][:B])
        ph[:B].copy_(pos, non_blocking=True)
        torch.cuda.current_stream(eng.device).synchronize()
        return ph[:B, : self.num_movables].numpy().copy()

    def count_achieved_goals(self, state: State) -> int:
        """puzzle.py:396-407 (a host-side comparison of caller-owned tuples; the batched
        engine computes the same count on the device for its reward)."""
        return sum(1 for s, g in zip(state[1 : 1 + len(self._goal_state)], self._goal_state) if tuple(s) == g)

    def is_goal_state(self, state: State) -> bool:
        """puzzle.py:409-411."""
        return _as_state(state[1 : 1 + len(self._goal_state)]) == self._goal_state

    def is_valid_plan(self, plan: Iterable[int]) -> bool:
        """puzzle.py:413-424: the whole plan is replayed on the device; a plan that reaches
        the goal before its last action is rejected like in the reference."""
        plan = [int(a) for a in plan]
        if self.is_goal_state(self._initial_state):
            return len(plan) == 0
        if not plan:
            return False
        if any(a not in (0, 1, 2, 3) for a in plan):
            raise ValueError("action must be one of 0 (L), 1 (R), 2 (U), 3 (D)")
        # ONE launch for the whole plan (pw_plan_states): the goal flag of every state it passes through
        _, goals = self._engine().plan_states(0, bytes(plan))
        return bool(goals[-1] == 1 and not goals[:-1].any())

    def expand4(self, states):
        """Planner successor expansion (cpp/include/search/best_first_search.h:76-78 calling
        PushWorldPuzzle::getNextState / satisfiesGoal, cpp/src/pushworld_puzzle.cc:386-469) for
        F states at once.

        Args:
            states: int32 array/tensor [F, N] of ``Position2D = x * 10000 + y`` values
                (pushworld_puzzle.h:32-37) in this puzzle's object order.

        Returns ``(succ int32 [F, 4, N], moved uint32-as-int32 [F, 4] bit masks of
        moved_object_indices, goal uint8 [F, 4])`` as torch tensors on the device.
        """
        eng = self._engine()
        st = torch.as_tensor(states, dtype=torch.int32).to(eng.device).contiguous()
        if st.dim() != 2 or st.shape[1] != self.num_movables:
            raise ValueError(f"states must have shape [F, {self.num_movables}]")
        F = st.shape[0]
        succ = torch.empty((F, 4, self.num_movables), dtype=torch.int32, device=eng.device)
        moved = torch.empty((F, 4), dtype=torch.int32, device=eng.device)
        goal = torch.empty((F, 4), dtype=torch.uint8, device=eng.device)
        if F:
            eng.expand4(0, st, succ, moved, goal)
        return succ, moved, goal

    # ---------------------------------------------------------------- rendering
    def render(self, state: State, border_width: int = DEFAULT_BORDER_WIDTH,
               pixels_per_cell: int = DEFAULT_PIXELS_PER_CELL) -> np.ndarray:
        """puzzle.py:426-469 on the GPU; uint8 (height, width, 3)."""
        if border_width < 1:
            raise ValueError("border_width must be >= 1")
        if pixels_per_cell < 1 + 2 * border_width:
            raise ValueError("pixels_per_cell must be >= 1 + 2*border_width")
        # rendered straight into pinned host memory (the kernel writes over PCIe: no copy command)
        eng = self._engine(pixels_per_cell, border_width, _capi.OBS_U8)
        b = self._state_bufs(eng)
        self._upload(eng, state)
        key = ("obs_host", pixels_per_cell, border_width, _capi.OBS_U8)
        if key not in self._bufs:
            self._bufs[key] = eng.alloc_obs_host(1)
        storage, view = self._bufs[key]
        eng.render(b["pid"], b["pos"], storage)
        torch.cuda.current_stream(eng.device).synchronize()
        return view[0].numpy().copy()

    def render_plan(self, plan: Iterable[int], border_width: int = DEFAULT_BORDER_WIDTH,
                    pixels_per_cell: int = DEFAULT_PIXELS_PER_CELL) -> List[np.ndarray]:
        """puzzle.py:471-506: one ``pw_plan_states`` launch leaves every state of the plan in a device buffer, ONE batched
        ``pw_render`` draws them all, one copy brings the images to the host."""
        if border_width < 1:
            raise ValueError("border_width must be >= 1")
        if pixels_per_cell < 1 + 2 * border_width:
            raise ValueError("pixels_per_cell must be >= 1 + 2*border_width")
        plan = [int(a) for a in plan]
        if any(a not in (0, 1, 2, 3) for a in plan):
            raise ValueError("action must be one of 0 (L), 1 (R), 2 (U), 3 (D)")
        eng = self._engine(pixels_per_cell, border_width, _capi.OBS_U8)
        n = len(plan) + 1
        dev_states = torch.zeros((n, eng.np, 2), dtype=torch.int8, device=eng.device)
        torch.cuda.current_stream(eng.device).synchronize()  # (the plan kernel runs on the engine's own stream)
        eng.plan_states(0, bytes(plan), dev_states=dev_states)
        pid = torch.zeros((n,), dtype=torch.int32, device=eng.device)
        storage, view = eng.alloc_obs(n)
        eng.render(pid, dev_states, storage)
        images = view.cpu().numpy()
        return [images[i] for i in range(n)]
