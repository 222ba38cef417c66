"""ctypes front-end of ``oracle/pw_oracle.c``  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The C file is the compiled restatement of the reference algorithm (tables + LIFO frontier +
painter) used as a fast checker in tests and as the ``cpu_baseline`` ("port") of bench.py.
Parsing comes from ``oracle/pw_oracle.py``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_int32, c_uint8, c_uint32, c_uint64, c_void_p

import numpy as np

from . import pw_oracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libpw_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "pw_oracle.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        l = ctypes.CDLL(_LIB)
        ip = POINTER(c_int)
        l.or_puzzle_create.restype = c_void_p
        l.or_puzzle_create.argtypes = [c_int, c_int, c_int, c_int, c_int, ip, ip, ip, ip, ip, ip, c_int, ip, c_int, ip]
        l.or_puzzle_destroy.argtypes = [c_void_p]
        l.or_static_size.argtypes = [c_void_p, c_int, c_int]
        l.or_dynamic_size.argtypes = [c_void_p, c_int, c_int, c_int]
        l.or_step.restype = c_uint32
        l.or_step.argtypes = [c_void_p, ip, c_int]
        l.or_count_goals.argtypes = [c_void_p, ip]
        l.or_env_step.argtypes = [c_void_p, ip, c_int, POINTER(c_double)]
        l.or_render.argtypes = [c_void_p, ip, c_int, c_int, c_void_p]
        l.or_observation_u8.argtypes = [c_void_p, ip, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
        l.or_observation_f32.argtypes = [c_void_p, ip, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
        l.or_rollout.restype = c_uint64
        l.or_rollout.argtypes = [POINTER(c_void_p), POINTER(c_int32), c_int, c_int, POINTER(c_uint8), c_int, c_int,
                                 c_int, c_int, c_int, c_int, c_int, POINTER(c_int)]
        l.or_rollout_trace.restype = None
        l.or_rollout_trace.argtypes = [POINTER(c_void_p), POINTER(c_int32), c_int, c_int, POINTER(c_uint8), c_int, c_int,
                                       c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        l.or_rollout_digest.restype = None
        l.or_rollout_digest.argtypes = [POINTER(c_void_p), POINTER(c_int32), c_int, c_int, POINTER(c_uint8), c_int, c_int,
                                        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        l.or_observe_batch.restype = None
        l.or_observe_batch.argtypes = [POINTER(c_void_p), POINTER(c_int32), c_void_p, c_int, POINTER(c_int32), c_int,
                                       c_int, c_int, c_int, c_int, c_int, c_void_p]
        l.or_set_thread_cpus.restype = None
        l.or_set_thread_cpus.argtypes = [POINTER(c_int), c_int]
        l.or_expand4_batch.restype = None
        l.or_expand4_batch.argtypes = [c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p]
        _lib = l
    return _lib


def _ints(seq):
    arr = np.ascontiguousarray(np.asarray(list(seq), dtype=np.int32).reshape(-1))
    if arr.size == 0:
        arr = np.zeros((2,), np.int32)
    return arr, arr.ctypes.data_as(POINTER(c_int))


class COraclePuzzle:
    def __init__(self, text: str, order: str = "python"):
        self.py = pw_oracle.OraclePuzzle(text, order=order, build_tables=False)
        p = self.py
        flat = lambda cells: [v for c in sorted(cells) for v in c]  # noqa: E731
        keep = []

        def arr(seq):
            a, ptr = _ints(seq)
            keep.append(a)
            return ptr

        self.handle = lib().or_puzzle_create(
            p.width, p.height, p.num_movables, p.num_goals, int(p.has_agent_walls),
            arr([len(s) for s in p.shapes]), arr([v for s in p.shapes for v in flat(s)]),
            arr([v for xy in p.initial_state for v in xy]), arr([v for xy in p.goal_state for v in xy]),
            arr([len(s) for s in p.goal_shapes]), arr([v for s in p.goal_shapes for v in flat(s)]),
            len(p.wall_cells), arr(flat(p.wall_cells)),
            len(p.agent_wall_cells), arr(flat(p.agent_wall_cells)),
        )
        assert self.handle, "or_puzzle_create failed"
        self.width, self.height = p.width, p.height
        self.num_movables, self.num_goals = p.num_movables, p.num_goals
        self.initial_state = p.initial_state

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.or_puzzle_destroy(self.handle)
            self.handle = None

    def _state(self, state):
        a = np.ascontiguousarray(np.asarray(state, dtype=np.int32).reshape(-1))
        return a, a.ctypes.data_as(POINTER(c_int))

    def get_next_state_moved(self, state, action):
        a, ptr = self._state(state)
        mask = lib().or_step(self.handle, ptr, int(action))
        nxt = tuple((int(a[2 * k]), int(a[2 * k + 1])) for k in range(self.num_movables))
        return nxt, [k for k in range(self.num_movables) if (mask >> k) & 1]

    def get_next_state(self, state, action):
        return self.get_next_state_moved(state, action)[0]

    def env_step(self, state, action):
        a, ptr = self._state(state)
        r = c_double()
        term = lib().or_env_step(self.handle, ptr, int(action), ctypes.byref(r))
        nxt = tuple((int(a[2 * k]), int(a[2 * k + 1])) for k in range(self.num_movables))
        return nxt, r.value, bool(term)

    def table_sizes(self):
        n = self.num_movables
        s = np.array([[lib().or_static_size(self.handle, a, i) for i in range(n)] for a in range(4)])
        d = np.array([[[lib().or_dynamic_size(self.handle, a, i, j) for j in range(n)] for i in range(n)]
                      for a in range(4)])
        return s, d

    def render(self, state, border_width=2, pixels_per_cell=20):
        _, ptr = self._state(state)
        img = np.zeros((self.height * pixels_per_cell, self.width * pixels_per_cell, 3), np.uint8)
        lib().or_render(self.handle, ptr, pixels_per_cell, border_width, img.ctypes.data)
        return img

    def observation(self, state, max_cell_height, max_cell_width, pixels_per_cell=20, border_width=2, dtype="f32"):
        _, ptr = self._state(state)
        ppc = pixels_per_cell
        scratch = np.zeros((self.height * ppc * self.width * ppc * 3,), np.uint8)
        if dtype == "f32":
            out = np.zeros((max_cell_height * ppc, max_cell_width * ppc, 3), np.float32)
            lib().or_observation_f32(self.handle, ptr, max_cell_height, max_cell_width, ppc, border_width,
                                     out.ctypes.data, scratch.ctypes.data)
        else:
            out = np.zeros((max_cell_height * ppc, max_cell_width * ppc, 3), np.uint8)
            lib().or_observation_u8(self.handle, ptr, max_cell_height, max_cell_width, ppc, border_width,
                                    out.ctypes.data, scratch.ctypes.data)
        return out


def rollout_trace(puzzles, puzzle_ids, actions, max_steps, autoreset, np_pad):
    """Every step of every environment (OpenMP over environments), from the initial states: returns
    ``pos int8 [T, B, np_pad, 2], reward f64 [T, B], terminated u8 [T, B], truncated u8 [T, B], steps i32 [T, B]``
    with next-step autoreset as in ``pw_step(PW_STEP_AUTORESET)`` when ``autoreset``."""
    handles = (c_void_p * len(puzzles))(*[p.handle for p in puzzles])
    pid = np.ascontiguousarray(np.asarray(puzzle_ids, dtype=np.int32))
    acts = np.ascontiguousarray(np.asarray(actions, dtype=np.uint8))
    T, B = acts.shape
    pos = np.zeros((T, B, np_pad, 2), np.int8)
    reward = np.zeros((T, B), np.float64)
    term = np.zeros((T, B), np.uint8)
    trunc = np.zeros((T, B), np.uint8)
    steps = np.zeros((T, B), np.int32)
    lib().or_rollout_trace(handles, pid.ctypes.data_as(POINTER(c_int32)), B, T, acts.ctypes.data_as(POINTER(c_uint8)),
                           int(-1 if max_steps is None else max_steps), int(bool(autoreset)), int(np_pad),
                           pos.ctypes.data, reward.ctypes.data, term.ctypes.data, trunc.ctypes.data, steps.ctypes.data)
    return pos, reward, term, trunc, steps


def rollout_digest(puzzles, puzzle_ids, actions, max_steps, autoreset, np_pad, weights):
    """``rollout_trace`` for long runs of big batches: per step and environment a 64-bit linear digest of the position row
    (``sum_k row[k] * weights[k]`` mod 2^64, ``weights`` int64 [np_pad * 2]) instead of the row; the rows themselves only after
    the last step.  Returns ``digest int64 [T, B], reward f64 [T, B], terminated u8 [T, B], truncated u8 [T, B], steps i32
    [T, B], pos_last int8 [B, np_pad, 2]``."""
    handles = (c_void_p * len(puzzles))(*[p.handle for p in puzzles])
    pid = np.ascontiguousarray(np.asarray(puzzle_ids, dtype=np.int32))
    acts = np.ascontiguousarray(np.asarray(actions, dtype=np.uint8))
    w = np.ascontiguousarray(np.asarray(weights, dtype=np.int64))
    assert w.shape == (np_pad * 2,)
    T, B = acts.shape
    digest = np.zeros((T, B), np.int64)
    reward = np.zeros((T, B), np.float64)
    term = np.zeros((T, B), np.uint8)
    trunc = np.zeros((T, B), np.uint8)
    steps = np.zeros((T, B), np.int32)
    pos_last = np.zeros((B, np_pad, 2), np.int8)
    lib().or_rollout_digest(handles, pid.ctypes.data_as(POINTER(c_int32)), B, T, acts.ctypes.data_as(POINTER(c_uint8)),
                            int(-1 if max_steps is None else max_steps), int(bool(autoreset)), int(np_pad), w.ctypes.data,
                            digest.ctypes.data, reward.ctypes.data, term.ctypes.data, trunc.ctypes.data, steps.ctypes.data,
                            pos_last.ctypes.data)
    return digest, reward, term, trunc, steps, pos_last


def observe_batch(puzzles, puzzle_ids, pos, sel, pad_h, pad_w, ppc, bw, dtype="u8"):
    """Padded observations of the environments ``sel`` of a batch with positions ``pos`` int8 [B, NP, 2]."""
    handles = (c_void_p * len(puzzles))(*[p.handle for p in puzzles])
    pid = np.ascontiguousarray(np.asarray(puzzle_ids, dtype=np.int32))
    pos = np.ascontiguousarray(np.asarray(pos, dtype=np.int8))
    sel = np.ascontiguousarray(np.asarray(sel, dtype=np.int32))
    out = np.zeros((len(sel), pad_h * ppc, pad_w * ppc, 3), np.float32 if dtype == "f32" else np.uint8)
    lib().or_observe_batch(handles, pid.ctypes.data_as(POINTER(c_int32)), pos.ctypes.data, pos.shape[1],
                           sel.ctypes.data_as(POINTER(c_int32)), len(sel), pad_h, pad_w, ppc, bw, int(dtype == "f32"),
                           out.ctypes.data)
    return out


def expand4_batch(puzzle, states, out=None):
    """``states`` int32 [F, N] Position2D -> (succ int32 [F, 4, N], moved uint32 [F, 4], goal uint8 [F, 4]); ``out``: the
    three arrays of an earlier call, written in place (a timed loop must not pay for fresh pages every pass)."""
    states = np.ascontiguousarray(np.asarray(states, dtype=np.int32))
    F, N = states.shape
    assert N == puzzle.num_movables
    if out is not None:
        succ, moved, goal = out
        assert succ.shape == (F, 4, N) and moved.shape == (F, 4) and goal.shape == (F, 4)
    else:
        succ = np.zeros((F, 4, N), np.int32)
        moved = np.zeros((F, 4), np.uint32)
        goal = np.zeros((F, 4), np.uint8)
    lib().or_expand4_batch(puzzle.handle, states.ctypes.data, F, succ.ctypes.data, moved.ctypes.data, goal.ctypes.data)
    return succ, moved, goal


def rollout(puzzles, puzzle_ids, actions, max_steps, render, pad_h, pad_w, ppc, bw, threads=0):
    """Batched CPU baseline: ``actions`` uint8 [T][B]; ``threads`` <= 0 = all OpenMP threads; ``render`` False / 0 no
    observation, True / 1 / "u8" the padded uint8 observation every step, 2 / "f32" the float32 one.
    Returns (checksum, threads used)."""
    render = {"u8": 1, "f32": 2, "uint8": 1, "float32": 2}.get(render, render) if isinstance(render, str) else int(render)
    handles = (c_void_p * len(puzzles))(*[p.handle for p in puzzles])
    pid = np.ascontiguousarray(np.asarray(puzzle_ids, dtype=np.int32))
    acts = np.ascontiguousarray(np.asarray(actions, dtype=np.uint8))
    T, B = acts.shape
    used = c_int(0)
    chk = lib().or_rollout(handles, pid.ctypes.data_as(POINTER(c_int32)), B, T,
                           acts.ctypes.data_as(POINTER(c_uint8)), int(-1 if max_steps is None else max_steps), int(render),
                           pad_h, pad_w, ppc, bw, int(threads), ctypes.byref(used))
    return int(chk), used.value


def set_thread_cpus(cpus):
    """OpenMP thread t of the baseline runs (``rollout``, ``expand4_batch``) pins itself to ``cpus[t % len(cpus)]``; an empty
    list switches the pinning off.  The calling thread is thread 0: save and restore its own mask around the runs."""
    arr = (c_int * max(1, len(cpus)))(*cpus)
    lib().or_set_thread_cpus(arr, len(cpus))
