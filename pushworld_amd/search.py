"""Breadth-first search with the frontier and the closed set on the GPU.

The reference planner (cpp/include/search/best_first_search.h:72-93) pops one node at a time and
calls ``getNextState`` four times; ``BreadthFirstSearch`` expands a whole layer per call through the
``pw_search_*`` entry points.  State numbering is deterministic: it is the numbering of a sequential
FIFO search that tries the actions in the order LEFT, RIGHT, UP, DOWN, so plans are the first
shortest plan in that order.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi
from .puzzle import PushWorldPuzzle

POSITION_LIMIT = 10000  # pushworld_puzzle.h:37


class LayerInfo(tuple):
    """(depth, new_states, total_states, goal_index) of one ``expand`` call."""

    depth = property(lambda self: self[0])
    new_states = property(lambda self: self[1])
    total_states = property(lambda self: self[2])
    goal_index = property(lambda self: self[3])


class BreadthFirstSearch:
    """Layer-synchronous BFS over the states of one puzzle.

    Args:
        puzzle: a ``PushWorldPuzzle`` (object order as parsed: Python order by default).
        max_states: capacity of the state store (device memory ~ ``max_states * (2 N + 21)`` bytes).
        chunk: parents per expansion pass (default 2^20; tests use small values to force many passes per layer).
        novelty_width: 0 = breadth-first search; 1 or 2 = width-limited search IW(k): new states whose
            novelty (reference ``NoveltyHeuristic``, novelty.cc:30-77) exceeds the width are closed but
            never expanded.  Incomplete but usually far smaller; plans are no longer guaranteed shortest.
    """

    def __init__(self, puzzle: PushWorldPuzzle, max_states: int = 1 << 22, novelty_width: int = 0,
                 chunk: Optional[int] = None):
        self.puzzle = puzzle
        self._engine = puzzle._engine()
        self.device = self._engine.device
        self.num_objects = puzzle.num_movables
        self.max_states = int(max_states)
        h = ctypes.c_void_p()
        self.novelty_width = int(novelty_width)
        self._engine.set_option("search_chunk", 0 if chunk is None else int(chunk))
        try:
            _capi.check(_capi.lib.pw_search_create(self._engine.handle, int(getattr(puzzle, "puzzle_index", 0)),
                                                   self.max_states, self.novelty_width, ctypes.byref(h)))
        finally:
            self._engine.set_option("search_chunk", 0)
        self.handle = h
        self.total_states = 0
        self.layers: List[Tuple[int, int]] = []  # (first index, count) per depth
        self.goal_index = -1
        self.exhausted = False

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def begin(self, start: Optional[Sequence[Tuple[int, int]]] = None) -> None:
        """Starts a search from ``start`` (a reference-style state, default: the initial state)."""
        arr = None
        if start is not None:
            if len(start) != self.num_objects:
                raise ValueError("start must hold one (x, y) pair per movable")
            arr = (ctypes.c_int32 * self.num_objects)(*[int(x) * POSITION_LIMIT + int(y) for x, y in start])
        _capi.check(_capi.lib.pw_search_begin(self.handle, arr, self._stream()))
        self.total_states = 1
        self.layers = [(0, 1)]
        state0 = tuple(start) if start is not None else self.puzzle.initial_state
        self.goal_index = 0 if self.puzzle.is_goal_state(state0) else -1
        self.exhausted = False

    def expand(self) -> LayerInfo:
        """Expands the newest layer; raises ``ValueError`` when the store is full."""
        info = (ctypes.c_int64 * 4)()
        rc = _capi.lib.pw_search_expand(self.handle, info, self._stream())
        depth, new, total, goal = (int(v) for v in info)
        if rc == _capi.PW_ELIMIT and total > self.total_states:  # store full: the last layer is incomplete
            self.layers.append((total - new, new))
            self.total_states = total
            self.goal_index = goal
        _capi.check(rc)
        if new:
            self.layers.append((total - new, new))
        else:
            self.exhausted = True
        self.total_states = total
        self.goal_index = goal
        return LayerInfo((depth, new, total, goal))

    def states(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """int array [count, N, 2] of (x, y) positions of states ``first .. first + count - 1``."""
        count = self.total_states - first if count is None else count
        out = torch.empty((count, self.num_objects), dtype=torch.int32, device=self.device)
        _capi.check(_capi.lib.pw_search_read_states(self.handle, first, count, _capi._ptr(out), self._stream()))
        v = out.cpu().numpy()
        return np.stack([v // POSITION_LIMIT, v % POSITION_LIMIT], axis=-1)

    def links(self, first: int = 0, count: Optional[int] = None):
        """(parent int32 [count], action uint8 [count]); the start state has parent -1."""
        count = self.total_states - first if count is None else count
        par = torch.empty((count,), dtype=torch.int32, device=self.device)
        act = torch.empty((count,), dtype=torch.uint8, device=self.device)
        _capi.check(_capi.lib.pw_search_read_links(self.handle, first, count, _capi._ptr(par), _capi._ptr(act),
                                                   self._stream()))
        return par.cpu().numpy(), act.cpu().numpy()

    def pruned(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """bool [count]: states cut by the novelty width (closed, never expanded)."""
        count = self.total_states - first if count is None else count
        out = torch.empty((count,), dtype=torch.uint8, device=self.device)
        _capi.check(_capi.lib.pw_search_read_flags(self.handle, first, count, _capi._ptr(out), self._stream()))
        return out.cpu().numpy().astype(bool)

    def plan(self, index: int) -> List[int]:
        """Actions leading from the start state to state ``index``."""
        cap = 256
        while True:
            buf = (ctypes.c_uint8 * cap)()
            n = _capi.check(_capi.lib.pw_search_plan(self.handle, int(index), buf, cap, self._stream()))
            if n <= cap:
                return [int(buf[i]) for i in range(n)]
            cap = n

    def solve(self, max_depth: Optional[int] = None) -> Optional[List[int]]:
        """A shortest plan (first in L, R, U, D order), or None if the reachable space holds no goal
        state within ``max_depth``.  ``ValueError`` if ``max_states`` is exhausted first."""
        if not self.layers:
            self.begin()
        while self.goal_index < 0 and not self.exhausted:
            if max_depth is not None and len(self.layers) - 1 >= max_depth:
                return None
            self.expand()
        return self.plan(self.goal_index) if self.goal_index >= 0 else None

    def close(self) -> None:
        h = getattr(self, "handle", None)
        if h and _capi.lib is not None:
            _capi.lib.pw_search_destroy(h)
            self.handle = None

    __del__ = close


class SetPuzzle:
    """Puzzle ``index`` of a packed ``_capi.PuzzleSet`` as ``BreadthFirstSearch`` needs it (number of movables,
    initial state, goal test, a state-only engine shared by all puzzles of the set) -- for sets that never existed
    as text (``generate.generate_level0_set``).  Reads the packed header (csrc/pw_format.h)."""

    def __init__(self, pset: "_capi.PuzzleSet", index: int, engine: Optional["_capi.Engine"] = None):
        hdr = pset.headers()[320 * index:320 * (index + 1)]
        self.puzzle_index = int(index)
        self.num_movables, n_goals = hdr[6], hdr[7]
        i8 = np.frombuffer(hdr, dtype=np.int8)
        self.initial_state = tuple((int(i8[256 + 2 * j]), int(i8[257 + 2 * j])) for j in range(self.num_movables))
        self.goal_state = tuple((int(i8[192 + 2 * g]), int(i8[193 + 2 * g])) for g in range(n_goals))
        self._eng = engine if engine is not None else _capi.Engine(pset, None, 3, 1, _capi.OBS_U8)

    def _engine(self):
        return self._eng

    def is_goal_state(self, state) -> bool:
        return tuple(tuple(int(v) for v in p) for p in state[1:1 + len(self.goal_state)]) == self.goal_state


class NoveltyTables:
    """Batched ``NoveltyHeuristic`` (cpp/src/heuristics/novelty.cc:30-77): ``evaluate`` returns, for an
    array of states, the values the reference would return when fed the states one by one in order."""

    def __init__(self, state_size: int, width: int, height: int, device: Optional[int] = None):
        from .puzzle import default_device_index

        self.device_index = default_device_index() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.state_size = int(state_size)
        h = ctypes.c_void_p()
        _capi.check(_capi.lib.pw_novelty_create(self.device_index, self.state_size, int(width), int(height),
                                                ctypes.byref(h)))
        self.handle = h

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self) -> None:
        _capi.check(_capi.lib.pw_novelty_reset(self.handle, self._stream()))

    def evaluate(self, states: torch.Tensor, moved: torch.Tensor) -> torch.Tensor:
        """``states`` int32 [F, N] Position2D (x * 10000 + y), ``moved`` int32/uint32-bit masks [F];
        returns uint8 [F] novelties (1, 2 or 3)."""
        if states.dtype != torch.int32 or states.dim() != 2 or states.shape[1] != self.state_size or \
                not states.is_contiguous() or states.device != self.device:
            raise ValueError("states must be a contiguous int32 tensor [F, state_size] on the tables' device")
        if moved.shape != (states.shape[0],) or moved.dtype not in (torch.int32, torch.uint32) or moved.device != self.device:
            raise ValueError("moved must be an int32 tensor [F] of bit masks on the tables' device")
        out = torch.empty((states.shape[0],), dtype=torch.uint8, device=self.device)
        _capi.check(_capi.lib.pw_novelty_eval(self.handle, _capi._ptr(states), _capi._ptr(moved), _capi._ptr(out),
                                              states.shape[0], self._stream()))
        return out

    def close(self) -> None:
        h = getattr(self, "handle", None)
        if h and _capi.lib is not None:
            _capi.lib.pw_novelty_destroy(h)
            self.handle = None

    __del__ = close


VERDICT_UNSOLVABLE, VERDICT_SOLVED, VERDICT_UNKNOWN, VERDICT_NOT_SEARCHED = 0, 1, 2, 3


def search_batch(engine, puzzle_indices=None, max_states: int = 1 << 16, plan_cap: int = 0):
    """``pw_search_batch``: breadth-first search of MANY small puzzles of ``engine``'s set in one launch (persistent workgroups,
    the whole search loop inside the kernel).  Returns numpy arrays ``(verdict uint8 [n], plan_len int32 [n], num_states
    int32 [n])``: verdict 1 solved (``plan_len`` = length of a shortest plan), 0 unsolvable, 2 unknown (more than
    ``max_states`` states), 3 not searched (beyond 16 x 16 cells / 8 movables: use ``BreadthFirstSearch``).  With
    ``plan_cap`` > 0 a fourth value: the list of plans (lists of actions; None where there is none or it is longer)."""
    dev = engine.device
    if puzzle_indices is None:
        n = len(engine.pset)
        idx = None
    else:
        idx = torch.as_tensor(np.asarray(puzzle_indices), dtype=torch.int32).to(dev)
        n = int(idx.shape[0])
    verdict = torch.empty((n,), dtype=torch.uint8, device=dev)
    plan_len = torch.empty((n,), dtype=torch.int32, device=dev)
    states = torch.empty((n,), dtype=torch.int32, device=dev)
    plans = torch.zeros((n, plan_cap), dtype=torch.uint8, device=dev) if plan_cap > 0 else None
    if n:
        _capi.check(_capi.lib.pw_search_batch(engine.handle, _capi._ptr(idx), n, int(max_states), 0, _capi._ptr(verdict),
                                              _capi._ptr(plan_len), _capi._ptr(states), _capi._ptr(plans), int(plan_cap),
                                              engine._stream()))
    v, pl, ns = verdict.cpu().numpy(), plan_len.cpu().numpy(), states.cpu().numpy()
    if plan_cap <= 0:
        return v, pl, ns
    ph = plans.cpu().numpy()
    return v, pl, ns, [ph[i, :pl[i]].tolist() if v[i] == VERDICT_SOLVED and 0 <= pl[i] <= plan_cap else None for i in range(n)]


def shortest_plan(puzzle, max_states: int = 1 << 20, plan_cap: int = 4096):
    """``(plan, verdict)`` of ONE puzzle from a single launch (``pw_search_batch`` with n = 1): the whole breadth-first search
    runs inside the kernel, so a search of a few dozen states costs one launch and one readback instead of a handful of
    launches and a readback per layer.  ``plan`` is a shortest plan (list of actions) or None; verdict as ``search_batch``.
    Puzzles beyond the kernel's limits (verdict 3) and searches beyond ``max_states`` (2) are ``BreadthFirstSearch``'s."""
    eng = puzzle._engine()
    v, pl, ns, plans = search_batch(eng, [int(getattr(puzzle, "puzzle_index", 0))], max_states=max_states, plan_cap=plan_cap)
    return plans[0], int(v[0])
