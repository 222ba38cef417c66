"""ctypes binding of ``libpushworld_amd.so`` (C ABI declared in ``include/pushworld_amd.h``).

The library is the product: there is no Python or CPU fallback.  If it has not been
built (``python -c "import __graft_entry__ as g; g.build()"`` or
``python -m pushworld_amd.build``) importing this module raises ``ImportError``, and any
device operation without a visible MI355X raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

# torch first: it loads its bundled HIP runtime (SONAME libamdhip64.so.7); loading our
# library afterwards binds to that same runtime, so tensor.data_ptr() values and
# torch.cuda streams are valid inside the kernels' process-wide HIP context.
import numpy as np
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PUSHWORLD_AMD_LIB") or os.path.join(_HERE, "lib", "libpushworld_amd.so")

ABI_VERSION = 4
PW_OK = 0
PW_EINVAL, PW_EPARSE, PW_EGOAL, PW_ELIMIT, PW_EDEVICE, PW_ENOMEM, PW_EELEMENT = -1, -2, -3, -4, -5, -6, -7
ORDER_PYTHON, ORDER_CPP = 0, 1
PUZZLE_HEADER_BYTES = 320      # sizeof(PwPuzzleHeader), csrc/pw_format.h (asserted by pw_puzzleset_headers users)
PUZZLE_HEADER_N_OFFSET = 6     # offsetof(PwPuzzleHeader, N): uint32 base, then uint8 W, H, N, G
OBS_U8, OBS_F32 = 0, 1
PLAN_MAX_ACTIONS = 65536  # PW_PLAN_MAX_ACTIONS of include/pushworld_amd.h: actions per pw_plan_states launch
STEP_AUTORESET = 1


class PwPuzzleInfo(ctypes.Structure):
    _fields_ = [
        ("width", c_int32),
        ("height", c_int32),
        ("num_movables", c_int32),
        ("num_goals", c_int32),
        ("num_wall_cells", c_int32),
        ("num_agent_wall_cells", c_int32),
        ("has_agent_walls", c_int32),
        ("order", c_int32),
    ]


class PwGenConfig(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("min_size", c_int32), ("max_size", c_int32),
        ("min_walls", c_int32), ("max_walls", c_int32),
        ("min_obstacles", c_int32), ("max_obstacles", c_int32),
        ("min_goal_objects", c_int32), ("max_goal_objects", c_int32),
        ("complex_shapes", c_int32),
    ]


class PwEngineConfig(ctypes.Structure):
    _fields_ = [
        ("max_steps", c_int32),
        ("pixels_per_cell", c_int32),
        ("border_width", c_int32),
        ("obs_dtype", c_int32),
        ("pad_cell_height", c_int32),
        ("pad_cell_width", c_int32),
        ("max_batch", c_int32),
    ]


# name -> (restype, argtypes); also the list checked by tests/test_capi_symbols.py
SIGNATURES = {
    "pw_last_error": (c_char_p, []),
    "pw_abi_version": (c_int, []),
    "pw_device_count": (c_int, []),
    "pw_puzzle_parse": (c_int, [c_char_p, c_size_t, c_int, POINTER(c_void_p)]),
    "pw_puzzle_destroy": (None, [c_void_p]),
    "pw_puzzle_info": (c_int, [c_void_p, POINTER(PwPuzzleInfo)]),
    "pw_puzzle_initial_state": (c_int, [c_void_p, POINTER(c_int32)]),
    "pw_puzzle_goal_state": (c_int, [c_void_p, POINTER(c_int32)]),
    "pw_puzzle_object_cells": (c_int, [c_void_p, c_int, POINTER(c_int32), c_int]),
    "pw_puzzle_goal_cells": (c_int, [c_void_p, c_int, POINTER(c_int32), c_int]),
    "pw_puzzle_wall_cells": (c_int, [c_void_p, POINTER(c_int32), c_int]),
    "pw_puzzle_agent_wall_cells": (c_int, [c_void_p, POINTER(c_int32), c_int]),
    "pw_puzzle_object_name": (c_int, [c_void_p, c_int, c_char_p, c_int]),
    "pw_puzzleset_create": (c_int, [POINTER(c_void_p), c_int, c_int, POINTER(c_void_p)]),
    "pw_puzzleset_destroy": (None, [c_void_p]),
    "pw_puzzleset_size": (c_int, [c_void_p]),
    "pw_puzzleset_max_dims": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "pw_puzzleset_blob": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]),
    "pw_puzzleset_save": (c_int, [c_void_p, c_char_p]),
    "pw_puzzleset_load": (c_int, [c_char_p, c_int, POINTER(c_void_p)]),
    "pw_engine_create": (c_int, [c_void_p, POINTER(PwEngineConfig), POINTER(c_void_p)]),
    "pw_engine_destroy": (None, [c_void_p]),
    "pw_engine_npad": (c_int, [c_void_p]),
    "pw_engine_obs_shape": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "pw_engine_obs_bytes": (c_int64, [c_void_p]),
    "pw_engine_render_kernel": (c_int, [c_void_p, c_char_p, c_int]),
    "pw_engine_obs_stride": (c_int64, [c_void_p]),
    "pw_reset": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "pw_mix64": (ctypes.c_uint64, [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]),
    "pw_resample": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_uint64, c_void_p, c_int32, c_void_p],
    ),
    "pw_step": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
         c_uint32, c_void_p],
    ),
    "pw_rollout": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_int32, c_uint32, c_void_p],
    ),
    "pw_batch_bind": (c_int, [c_void_p, c_void_p, c_int32, POINTER(c_int64), c_void_p]),
    "pw_batch_unbind": (c_int, [c_void_p]),
    "pw_mailbox_open": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_uint32, c_int32, c_int32,
         POINTER(c_void_p)],
    ),
    "pw_mailbox_post": (c_int, [c_void_p, c_void_p, c_int32, POINTER(ctypes.c_uint64)]),
    "pw_mailbox_wait": (c_int, [c_void_p, ctypes.c_uint64, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "pw_mailbox_run": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, POINTER(ctypes.c_uint64)]),
    "pw_mailbox_close_profile": (c_int, [c_void_p, POINTER(c_int64)]),
    "pw_mailbox_step": (c_int, [c_void_p, c_void_p, c_int32, POINTER(ctypes.c_uint64)]),
    "pw_mailbox_layout": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int32)]),
    "pw_mailbox_close": (c_int, [c_void_p]),
    "pw_engine_set_step_signal": (c_int, [c_void_p, c_void_p]),
    "pw_engine_set_step_host_copy": (c_int, [c_void_p, c_void_p]),
    "pw_render": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "pw_step_render": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_int64, c_int32, c_uint32, c_void_p],
    ),
    "pw_novelty_create": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "pw_novelty_destroy": (None, [c_void_p]),
    "pw_novelty_reset": (c_int, [c_void_p, c_void_p]),
    "pw_novelty_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "pw_search_create": (c_int, [c_void_p, c_int32, c_int64, c_int32, POINTER(c_void_p)]),
    "pw_search_read_flags": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "pw_search_destroy": (None, [c_void_p]),
    "pw_search_begin": (c_int, [c_void_p, POINTER(c_int32), c_void_p]),
    "pw_search_expand": (c_int, [c_void_p, POINTER(c_int64), c_void_p]),
    "pw_search_read_states": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "pw_search_read_links": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "pw_search_plan": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    "pw_step_render_delta": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_int64, c_int32, c_uint32, c_void_p],
    ),
    "pw_expand4": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "pw_generate_level0": (c_int, [c_int, POINTER(PwGenConfig), ctypes.c_uint64, c_int32, c_void_p, c_void_p, c_void_p]),
    "pw_generate_level0_attempts": (c_int, [POINTER(PwGenConfig), ctypes.c_uint64, c_int32, c_void_p]),
    "pw_transform_grids": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "pw_puzzleset_from_grids": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int, POINTER(c_void_p), c_void_p]),
    "pw_grid_to_text": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_char_p, c_int32]),
    "pw_puzzleset_headers": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]),
    "pw_engine_set_option": (c_int, [c_void_p, c_int32, c_int64]),
    "pw_engine_get_option": (c_int64, [c_void_p, c_int32]),
    "pw_engine_profile_read": (c_int, [c_void_p, POINTER(ctypes.c_float), c_int32]),
    "pw_engine_tune_render": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "pw_engine_bad_actions": (c_int64, [c_void_p, c_void_p]),
    "pw_validate_state": (c_int64, [c_void_p, c_void_p, c_void_p, c_int32, POINTER(c_int32), c_void_p]),
    "pw_puzzleset_overlap_tables": (c_int64, [c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "pw_obs_alloc": (c_int, [c_void_p, c_int32, POINTER(c_void_p)]),
    "pw_obs_alloc_tuned": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, POINTER(c_void_p),
                                   POINTER(ctypes.c_float), POINTER(c_int32), c_void_p]),
    "pw_obs_free": (c_int, [c_void_p, c_void_p]),
    "pw_search_batch": (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                c_void_p]),
    "pw_counters": (c_int, [c_void_p, POINTER(c_int64), c_void_p]),
    "pw_counters_reset": (c_int, [c_void_p, c_void_p]),
    "pw_next_state": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "pw_plan_states": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
}

# pw_engine_set_option keys (include/pushworld_amd.h)
OPTIONS = {
    "step_kernel": 1,        # 0 / "group", 1 / "wave", 2 / "lane"
    "fused_step_render": 2,
    "render_kernel": 3,      # 0 / "auto", 1 / "lds"
    "page_slice_envs": 4,
    "search_chunk": 5,
    "profile_render": 6,
    "page_order": 7,
    "page_run_log2": 8,
    "page_lds_pad_kb": 9,
    "step_lds_tables": 10,
    "tuned_ns": 11,          # read-only
    "step_wide_groups": 12,  # N_pad 32: 0 two movables per lane (16-lane groups), 1 32-lane groups
    "page_load_all": 13,     # ppc-3 page kernel: every page loads its static chunks
    "obs_chunk_mb": 14,      # pw_obs_alloc: MiB per physical chunk (0 = default, 32)
    "obs_accept_gbs": 15,    # pw_obs_alloc_tuned: rate at which a candidate buffer is kept right away
    "step_tables": 16,       # overlap tables: 0 / "auto" = 1 / "all" (every puzzle), 2 / "none", 3 / "big" (movables beyond 8 x 8 only)
    "step_table_bytes": 17,    # read-only
    "step_table_puzzles": 18,  # read-only
    "step_narrow_groups": 19,  # N_pad 16: 8-lane groups with two movables per lane
    "step_block_order": 20,    # 0 / "forward", 1 / "reverse": which end of the batch the step kernel starts with
    "step_boards": 22,         # sets of 8 x 8 puzzles, state only: 0 / "auto" whole-grid boards in registers, 2 / "never"
    "step_board_set": 23,      # read-only: the set qualifies
    "expand_lds_tables": 24,   # pw_expand4, one lane per state: 0 / "auto" push tables in LDS where they fit, 2 / "never", 3 never + whole runs
    "expand_tile_order": 25,   # pw_expand4_v2_kernel: 0 tiles interleaved, 1 a contiguous eighth of the frontier per XCD
    "expand_prefetch": 26,     # ... 1 = next tile's rows in flight while this one is computed
    "step_mixed_groups": 30,  # N_pad 8 / 16 sets with tables: lanes per environment chosen per workgroup (0 auto, 2 never)
    "expand_wg_waves": 29,  # pw_expand4 with one (large-table) workgroup per CU: cap on its wavefronts (0 automatic)
    "search_batch_groups_per_cu": 28,  # pw_search_batch: persistent workgroups per CU (0 = automatic)
    "expand_groups_per_cu": 27,  # ... persistent workgroups per CU (0 = automatic)
    "step_quad16": 31,         # 16 x 16 whole-grid boards, four lanes per environment: 0 / "auto", 2 / "never"
    "search_keys": 33,         # closed set of the searches created afterwards: 0 / "fingerprint" (default), 1 / "exact" 63-bit keys where they fit
    "expand_pair_dims": 34,    # pw_expand4 (tables in LDS): 0 / "auto" pair tables sized per pair where the uniform ones exceed 16 KB, 2 / "never"
    "step_quad16_puzzles": 32, # read-only: puzzles of the set that fit
    "mailbox_mode": 35,        # pw_mailbox_open: 0 / 1 / 2 / 3 who polls the host's word, + 4 fences, 11 = 3 pipelined (default), + 16 per-phase clock
    "step_lane_batch": 21,     # state-only launches of >= this many environments: one lane per environment (0 default, "never")
    "obs_tune_ms": 40,         # pw_obs_alloc_tuned: wall-clock budget of the candidate screen (0 = default 10 000 ms)
    "obs_screen_ms": 41,       # read-only: what the last screen took
    "step_one_applies": 49,    # read-only: step_render_delta on a batch of one with a completion word is ONE launch on this engine
    "mailbox_form": 48,        # read-only: 0 no mailbox open, 1 lanes / boards, 2 the segments of the bound batch
    "mailbox_seg": 47,         # the resident kernel of a fully bound batch runs its segments (tables in LDS): 0 automatic, 2 never
    "step_one_fused": 46,      # step_render_delta on a batch of one with a completion word: 1 (default) one launch, 2 ... writing whole rows, 0 two launches
    "bind_min_envs": 36,       # pw_batch_bind: environments of a batch that must play a puzzle for it to be bound (0 = default 48)
    "bind_fused": 37,          # launches of a bound batch: 0 / "auto" one launch for segments + lane groups, 2 two launches
    "bind_max_kb": 44,         # largest table block (KiB of LDS) that is bound (0 = default 48)
    "bind_rollouts": 43,       # pw_rollout on a bound batch: 0 / "auto" segments when every environment is bound, 1 always, 2 never
    "bind_spread": 45,         # environments per wavefront of the segment kernels: 0 automatic, 1 .. 5 = at most 64 / 32 / 16 / 8 / 4 (+ 16 x the same: listed lanes only)
    "bind_lanes": 42,          # lanes per environment of the segments: 0 automatic, 1 / 2 / 3 / 4 = at most 1 / 2 / 4 / 8
    "bind_puzzles": 38,        # read-only: puzzles of the set that can be bound (table block within 16 KB of LDS)
    "bind_mismatches": 39,     # read-only: environments found with another puzzle id than the one they were bound to
}
_OPTION_VALUES = {"group": 0, "wave": 1, "lane": 2, "auto": 0, "page": 0, "lds": 1, "big": 3, "all": 1, "none": 2,
                  "forward": 0, "reverse": 1, "never": 2**31}


# names that mean different numbers for different options ("never" is a batch threshold for step_lane_batch)
_OPTION_VALUES_BY_KEY = {"step_boards": {"auto": 0, "never": 2},
                         "step_quad16": {"auto": 0, "never": 2},
                         "search_keys": {"fingerprint": 0, "exact": 1},
                         "expand_pair_dims": {"auto": 0, "never": 2},
                         "step_narrow_groups": {"auto": 0, "always": 1, "never": 2},
                         "expand_lds_tables": {"auto": 0, "never": 2},
                         "step_mixed_groups": {"auto": 0, "never": 2},
                         "step_lds_tables": {"auto": 0, "always": 1, "never": 2}}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -m pushworld_amd.build).  There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.pw_abi_version() != ABI_VERSION:
        raise ImportError("libpushworld_amd.so ABI version mismatch; rebuild it")
    return lib


lib = _load()


def last_error() -> str:
    msg = lib.pw_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> int:
    """Maps error codes to the exception types of the reference."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc in (PW_EINVAL, PW_EPARSE, PW_ELIMIT):
        raise ValueError(msg)
    if rc == PW_EGOAL:
        raise AssertionError(msg)
    if rc == PW_ENOMEM:
        raise MemoryError(msg)
    if rc == PW_EELEMENT:
        raise IndexError(msg)
    raise RuntimeError(msg)


def device_count() -> int:
    n = lib.pw_device_count()
    return n if n > 0 else 0


def _cells(fn, handle, *idx):
    n = check(fn(handle, *idx, None, 0))
    buf = (c_int32 * (2 * max(n, 1)))()
    check(fn(handle, *idx, buf, n))
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]


class ParsedPuzzle:
    """Owner of a ``PwPuzzle*`` (host-side parse products)."""

    def __init__(self, text: str, order: int = ORDER_PYTHON):
        data = text.encode("utf-8")
        h = c_void_p()
        check(lib.pw_puzzle_parse(data, len(data), order, ctypes.byref(h)))
        self.handle = h
        info = PwPuzzleInfo()
        check(lib.pw_puzzle_info(h, ctypes.byref(info)))
        self.width, self.height = info.width, info.height
        self.num_movables, self.num_goals = info.num_movables, info.num_goals
        self.has_agent_walls = bool(info.has_agent_walls)
        self.order = info.order
        n, g = self.num_movables, self.num_goals
        buf = (c_int32 * (2 * n))()
        check(lib.pw_puzzle_initial_state(h, buf))
        self.initial_state = tuple((buf[2 * i], buf[2 * i + 1]) for i in range(n))
        gbuf = (c_int32 * (2 * max(g, 1)))()
        check(lib.pw_puzzle_goal_state(h, gbuf))
        self.goal_state = tuple((gbuf[2 * i], gbuf[2 * i + 1]) for i in range(g))
        self._cache = {}

    # cell lists are only needed by the Python mirrors' properties: fetched on first use so that
    # loading the 15 400 level-0 puzzles stays cheap
    def _lazy(self, key, make):
        if key not in self._cache:
            self._cache[key] = make()
        return self._cache[key]

    @property
    def object_cells(self):
        return self._lazy("obj", lambda: [_cells(lib.pw_puzzle_object_cells, self.handle, j)
                                          for j in range(self.num_movables)])

    @property
    def goal_cells(self):
        return self._lazy("goal", lambda: [_cells(lib.pw_puzzle_goal_cells, self.handle, k)
                                           for k in range(self.num_goals)])

    @property
    def wall_cells(self):
        return self._lazy("wall", lambda: _cells(lib.pw_puzzle_wall_cells, self.handle))

    @property
    def agent_wall_cells(self):
        return self._lazy("aw", lambda: _cells(lib.pw_puzzle_agent_wall_cells, self.handle))

    @property
    def names(self):
        def make():
            out = []
            for j in range(self.num_movables):
                nb = ctypes.create_string_buffer(64)
                check(lib.pw_puzzle_object_name(self.handle, j, nb, 64))
                out.append(nb.value.decode())
            return out

        return self._lazy("names", make)

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and lib is not None:
            lib.pw_puzzle_destroy(h)
            self.handle = None


class PuzzleSet:
    """Owner of a ``PwPuzzleSet*``: packed tables, uploaded to ``device`` (>= 0)."""

    def __init__(self, puzzles, device: int):
        self.puzzles = list(puzzles)
        arr = (c_void_p * len(self.puzzles))(*[p.handle for p in self.puzzles])
        h = c_void_p()
        check(lib.pw_puzzleset_create(arr, len(self.puzzles), device, ctypes.byref(h)))
        self.handle = h
        self.device = device
        self.count = len(self.puzzles)
        self._read_dims()

    def _read_dims(self):
        w, hh, n = c_int(), c_int(), c_int()
        check(lib.pw_puzzleset_max_dims(self.handle, ctypes.byref(w), ctypes.byref(hh), ctypes.byref(n)))
        self.max_width, self.max_height, self.max_movables = w.value, hh.value, n.value

    def __len__(self):
        return self.count

    def save(self, path: str) -> None:
        """Writes the packed set (``pw_puzzleset_save``)."""
        check(lib.pw_puzzleset_save(self.handle, os.fsencode(path)))

    @classmethod
    def load(cls, path: str, device: int) -> "PuzzleSet":
        """A set from a packed file (``pw_puzzleset_load``); ``puzzles`` is None (no parsed texts)."""
        self = cls.__new__(cls)
        self.puzzles = None
        h = c_void_p()
        check(lib.pw_puzzleset_load(os.fsencode(path), device, ctypes.byref(h)))
        self.handle = h
        self.device = device
        self.count = lib.pw_puzzleset_size(h)
        self._read_dims()
        return self

    @classmethod
    def from_grids(cls, grids, dims, device: int, order: int = ORDER_PYTHON) -> "PuzzleSet":
        """A set packed ON THE DEVICE from symbol grids (``pw_puzzleset_from_grids``): ``grids`` uint8 tensor
        [count, slot_w * slot_w], ``dims`` int32 tensor [count, 2] on ``device``; ``puzzles`` is None."""
        self = cls.__new__(cls)
        self.puzzles = None
        slot_w = int(round(grids.shape[-1] ** 0.5))
        if slot_w * slot_w != grids.shape[-1] or dims.shape != (grids.shape[0], 2):
            raise ValueError("grids must be [count, slot_w * slot_w] with dims [count, 2]")
        h = c_void_p()
        stream = c_void_p(torch.cuda.current_stream(torch.device("cuda", device)).cuda_stream)
        check(lib.pw_puzzleset_from_grids(device, _ptr(grids), _ptr(dims), grids.shape[0], slot_w, order, ctypes.byref(h),
                                          stream))
        self.handle = h
        self.device = device
        self.count = lib.pw_puzzleset_size(h)
        self._read_dims()
        return self

    def blob(self) -> bytes:
        p, n = c_void_p(), c_size_t()
        check(lib.pw_puzzleset_blob(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p.value, n.value)

    def overlap_tables(self, mode: int = 0):
        """``pw_puzzleset_overlap_tables``: (words uint64 [n], dir uint32 [count, 4]) as numpy arrays."""
        import numpy as np

        n = check(lib.pw_puzzleset_overlap_tables(self.handle, mode, None, 0, None))
        words = np.zeros((n,), np.uint64)
        d = np.zeros((self.count, 4), np.uint32)
        check(lib.pw_puzzleset_overlap_tables(self.handle, mode, c_void_p(words.ctypes.data), n, c_void_p(d.ctypes.data)))
        return words, d

    def headers(self) -> bytes:
        """The packed ``PwPuzzleHeader`` array (320 bytes per puzzle, csrc/pw_format.h)."""
        p, n = c_void_p(), c_size_t()
        check(lib.pw_puzzleset_headers(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p.value, n.value)

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and lib is not None:
            lib.pw_puzzleset_destroy(h)
            self.handle = None


def _ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


# the raw hipStream_t of torch's current stream without building a torch.cuda.Stream object (~0.2 us instead of ~1.5 us;
# the step entry points are called once per environment step)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if _raw_stream is None:  # pragma: no cover -- older torch
    def _raw_stream(device_index):
        return torch.cuda.current_stream(device_index).cuda_stream


class Mailbox:
    """``pw_mailbox_*`` (include/pushworld_amd.h): a resident kernel that steps a small state-only batch whenever the host posts
    the actions of a step -- no launch, no stream synchronisation.  ``post`` returns the step's number, ``wait`` the verdicts of
    that step as numpy views of pinned host memory (valid until ``ring`` more steps have been posted), ``step`` does both.
    The arrays given here receive after every step what ``pw_step`` would have written.  A context manager: the kernel ends at
    ``close``, or by itself after ``idle_ms`` without a post (a resident kernel would hold every device-wide synchronisation --
    ``torch.cuda.synchronize``, a ``hipFree`` of torch's allocator, destroying another engine -- for ever; such a call made
    while a mailbox is open costs up to ``idle_ms`` and ends the kernel).  A ``post`` / ``step`` / ``run`` that finds the mailbox
    expired with no step in flight opens it again by itself (``reopened`` counts it: the arrays hold the state after the last
    complete step); with steps in flight it raises RuntimeError.  For loops that pause between steps (learner updates) raise
    ``idle_ms``: a pause shorter than ``idle_ms / 2`` costs nothing."""

    def __init__(self, engine, puzzle_id, pos, steps, reward, dgoals, terminated, truncated, flags=0, ring=8, idle_ms=1000):
        self.engine = engine
        self.batch = int(pos.shape[0])
        self._keep = (puzzle_id, pos, steps, reward, dgoals, terminated, truncated)
        self._open_args = (int(flags), int(ring), int(idle_ms))
        self.reopened = 0  # how often an expired mailbox was opened again (see _revive)
        self.handle = None
        self._open()

    def _open(self):
        engine = self.engine
        puzzle_id, pos, steps, reward, dgoals, terminated, truncated = self._keep
        flags, ring, idle_ms = self._open_args
        torch.cuda.current_stream(engine.device).synchronize()  # what was queued for these arrays is complete
        handle = c_void_p()
        check(lib.pw_mailbox_open(engine.handle, _ptr(puzzle_id), _ptr(pos), _ptr(steps), _ptr(reward), _ptr(dgoals),
                                  _ptr(terminated), _ptr(truncated), self.batch, flags, ring, idle_ms, ctypes.byref(handle)))
        self.handle = handle
        import weakref
        engine._mailbox = weakref.ref(self)
        self._seq = ctypes.c_uint64()
        self._seq_ref = ctypes.byref(self._seq)
        self._waited = 0  # the highest step number a wait has returned (steps posted beyond it are in flight)
        self._out = (c_void_p(), c_void_p(), c_void_p())
        # numpy views of the pinned result slots, made once (a view per call costs more than the step)
        base, stride, ot, ou, rg = c_void_p(), c_int64(), c_int64(), c_int64(), c_int32()
        check(lib.pw_mailbox_layout(handle, ctypes.byref(base), ctypes.byref(stride), ctypes.byref(ot), ctypes.byref(ou), ctypes.byref(rg)))
        self.ring = rg.value
        B = self.batch
        raw = np.ctypeslib.as_array(ctypes.cast(base, POINTER(ctypes.c_uint8)), (self.ring * stride.value,))
        self._slots = []
        for i in range(self.ring):
            s = raw[i * stride.value:(i + 1) * stride.value]
            self._slots.append((s[:8 * B].view(np.float64), s[ot.value:ot.value + B], s[ou.value:ou.value + B]))

    def _revive(self, rc) -> bool:
        """A post that failed because the mailbox EXPIRED -- the host paused for more than ``idle_ms / 2`` (a learner update, logging,
        garbage collection), or a device-wide synchronisation made the resident kernel run into its idle limit -- with no step in
        flight: the arrays hold the state after the last complete step, so the mailbox is closed and opened again and the post
        retried.  (Step numbers restart at 1; with steps in flight the caller still holds numbers of the old kernel: the error is
        raised as before.)"""
        if rc != PW_EDEVICE or self._seq.value != self._waited or b"expired" not in (lib.pw_last_error() or b""):
            return False
        handle, self.handle = self.handle, None
        lib.pw_mailbox_close(handle)
        self._open()
        self.reopened += 1
        return True

    def post(self, actions) -> int:
        """``actions``: uint8 [B], a numpy array (host) or a tensor on the engine's device (torch's current stream is
        synchronised first: the resident kernel is not in any stream's order)."""
        if self.handle is None:
            raise RuntimeError("the mailbox is closed")
        if isinstance(actions, np.ndarray):
            if actions.dtype != np.uint8 or actions.shape != (self.batch,) or not actions.flags.c_contiguous:
                raise ValueError("actions must be a contiguous uint8 array of shape [num_envs]")
            rc = lib.pw_mailbox_post(self.handle, c_void_p(actions.ctypes.data), 1, self._seq_ref)
            if rc and self._revive(rc):
                rc = lib.pw_mailbox_post(self.handle, c_void_p(actions.ctypes.data), 1, self._seq_ref)
            check(rc)
        else:
            if actions.dtype != torch.uint8 or tuple(actions.shape) != (self.batch,) or not actions.is_contiguous() \
                    or actions.device != self.engine.device:
                raise ValueError("actions must be a contiguous uint8 tensor of shape [num_envs] on the engine's device")
            torch.cuda.current_stream(self.engine.device).synchronize()  # (the kernel that produced them has finished)
            rc = lib.pw_mailbox_post(self.handle, _ptr(actions), 0, self._seq_ref)
            if rc and self._revive(rc):
                rc = lib.pw_mailbox_post(self.handle, _ptr(actions), 0, self._seq_ref)
            check(rc)
        return self._seq.value

    def wait(self, seq: int):
        if self.handle is None:
            raise RuntimeError("the mailbox is closed")
        check(lib.pw_mailbox_wait(self.handle, seq, None, None, None))
        self._waited = max(self._waited, int(seq))
        return self._slots[(seq - 1) % self.ring]

    def step(self, actions):
        """``post`` + ``wait`` in one foreign call: ``(reward, terminated, truncated)`` of this step."""
        if self.handle is None:
            raise RuntimeError("the mailbox is closed")
        if type(actions) is np.ndarray:
            if actions.dtype != np.uint8 or actions.shape != (self.batch,) or not actions.flags.c_contiguous:
                raise ValueError("actions must be a contiguous uint8 array of shape [num_envs]")
            rc = lib.pw_mailbox_step(self.handle, actions.ctypes.data, 1, self._seq_ref)
            if rc and self._revive(rc):
                rc = lib.pw_mailbox_step(self.handle, actions.ctypes.data, 1, self._seq_ref)
        else:
            if actions.dtype != torch.uint8 or tuple(actions.shape) != (self.batch,) or not actions.is_contiguous() \
                    or actions.device != self.engine.device:
                raise ValueError("actions must be a contiguous uint8 tensor of shape [num_envs] on the engine's device")
            torch.cuda.current_stream(self.engine.device).synchronize()  # (the kernel that produced them has finished)
            rc = lib.pw_mailbox_step(self.handle, actions.data_ptr(), 0, self._seq_ref)
            if rc and self._revive(rc):
                rc = lib.pw_mailbox_step(self.handle, actions.data_ptr(), 0, self._seq_ref)
        if rc:
            check(rc)
        self._waited = self._seq.value
        return self._slots[(self._seq.value - 1) % self.ring]

    def run(self, actions, ahead: int = 1) -> int:
        """``pw_mailbox_run``: the steps of a uint8 [T, B] array (numpy: host, tensor: device) with at most ``ahead`` in flight;
        returns the number of the last step (``wait`` gives the verdicts of the last ``ring`` steps)."""
        if self.handle is None:
            raise RuntimeError("the mailbox is closed")
        host = isinstance(actions, np.ndarray)
        if host:
            ok = actions.dtype == np.uint8 and actions.ndim == 2 and actions.shape[1] == self.batch and actions.flags.c_contiguous
            ptr = c_void_p(actions.ctypes.data)
        else:
            ok = actions.dtype == torch.uint8 and actions.dim() == 2 and actions.shape[1] == self.batch and actions.is_contiguous() \
                and actions.device == self.engine.device
            ptr = _ptr(actions)
        if not ok:
            raise ValueError("actions must be a contiguous uint8 array / tensor of shape [T, num_envs]")
        rc = lib.pw_mailbox_run(self.handle, ptr, int(actions.shape[0]), 1 if host else 0, int(ahead), self._seq_ref)
        if rc and self._revive(rc):
            rc = lib.pw_mailbox_run(self.handle, ptr, int(actions.shape[0]), 1 if host else 0, int(ahead), self._seq_ref)
        check(rc)
        self._waited = self._seq.value
        return self._seq.value

    def close(self, profile: bool = False):
        """Ends the kernel.  ``profile=True``: returns microseconds per step that wavefront 0 spent waiting for the host's word,
        reading its actions, stepping, storing, and counting itself in (``pw_mailbox_close_profile``; the engine option
        ``mailbox_mode`` needs bit 4 (+16) set before the mailbox is opened: the clock is off by default)."""
        if self.handle is None:
            return None
        handle, self.handle = self.handle, None
        if not profile:
            check(lib.pw_mailbox_close(handle))
            return None
        raw = (c_int64 * 8)()
        check(lib.pw_mailbox_close_profile(handle, raw))
        out = {"steps": raw[0], "ended_by": {2: "stop", 3: "idle"}.get(raw[1], raw[1])}
        if raw[0] > 0 and raw[7] > 0:
            for i, n in enumerate(("wait_for_word", "read_actions", "step", "store", "arrive")):
                out[n + "_us"] = round(raw[2 + i] / raw[7] * 1e3 / raw[0], 3)
        return out

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class _DLDevice(ctypes.Structure):
    _fields_ = [("device_type", c_int), ("device_id", c_int)]


class _DLDataType(ctypes.Structure):
    _fields_ = [("code", ctypes.c_uint8), ("bits", ctypes.c_uint8), ("lanes", ctypes.c_uint16)]


class _DLTensor(ctypes.Structure):
    _fields_ = [("data", c_void_p), ("device", _DLDevice), ("ndim", c_int), ("dtype", _DLDataType),
                ("shape", POINTER(c_int64)), ("strides", POINTER(c_int64)), ("byte_offset", ctypes.c_uint64)]


class _DLManagedTensor(ctypes.Structure):
    pass


_DL_DELETER = ctypes.CFUNCTYPE(None, POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", c_void_p), ("deleter", _DL_DELETER)]
_dl_live = {}  # address of a DLManagedTensor -> (struct, shape array, holder): alive until torch calls the deleter


@_DL_DELETER
def _dl_deleter(mt):
    _dl_live.pop(ctypes.addressof(mt.contents), None)  # drops the holder: its __del__ frees the buffer


def _dlpack_tensor(holder, device_index: int, esz: int):
    """A torch tensor over ``holder.ptr`` through a DLPack capsule (kDLROCM): no pointer-attribute query, torch calls
    the deleter when the last tensor over the memory has gone."""
    shape = (c_int64 * len(holder.shape))(*holder.shape)
    mt = _DLManagedTensor()
    mt.dl_tensor.data = holder.ptr
    mt.dl_tensor.device = _DLDevice(10, device_index)  # kDLROCM
    mt.dl_tensor.ndim = len(holder.shape)
    mt.dl_tensor.dtype = _DLDataType(1 if esz == 1 else 2, 8 * esz, 1)  # kDLUInt / kDLFloat
    mt.dl_tensor.shape = shape
    mt.dl_tensor.strides = None
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _dl_deleter
    key = ctypes.addressof(mt)
    _dl_live[key] = (mt, shape, holder)
    new_capsule = ctypes.pythonapi.PyCapsule_New
    new_capsule.restype = ctypes.py_object
    new_capsule.argtypes = [c_void_p, c_char_p, c_void_p]
    try:
        return torch.from_dlpack(new_capsule(key, b"dltensor", None))
    except Exception:
        _dl_live.pop(key, None)
        raise


class _OwnedObs:
    """Keeps a ``pw_obs_alloc`` buffer alive for the tensors over it (``__cuda_array_interface__``: torch holds a
    reference to this object for as long as any tensor shares the memory) and frees it afterwards."""

    def __init__(self, engine, ptr, shape, typestr):
        self.engine = engine  # the engine must outlive its buffers
        self.ptr = ptr
        self.shape = tuple(shape)
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 3,
                                         "strides": None}

    def __del__(self):
        eng, ptr = getattr(self, "engine", None), getattr(self, "ptr", None)
        if eng is not None and ptr and getattr(eng, "handle", None) and lib is not None:
            lib.pw_obs_free(eng.handle, c_void_p(ptr))
        self.ptr = None


class Engine:
    """Owner of a ``PwEngine*``.  Methods take torch tensors resident on the set's device and
    enqueue kernels on ``torch.cuda.current_stream()``."""

    def __init__(self, pset: PuzzleSet, max_steps=None, pixels_per_cell=20, border_width=2,
                 obs_dtype=OBS_F32, pad_cell_height=0, pad_cell_width=0, options=None, max_batch=0):
        if max_steps is not None and int(max_steps) < 0:
            raise ValueError("max_steps must be None or >= 0")
        cfg = PwEngineConfig(
            # None = never truncate; 0 truncates on every step, like `steps >= max_steps` in gym_env.py:223
            int(max_steps) if max_steps is not None else -1,
            int(pixels_per_cell), int(border_width), int(obs_dtype),
            int(pad_cell_height), int(pad_cell_width),
            # engine-owned per-environment scratch sized at creation: no entry point allocates afterwards
            int(max_batch),
        )
        h = c_void_p()
        check(lib.pw_engine_create(pset.handle, ctypes.byref(cfg), ctypes.byref(h)))
        self.handle = h
        self.pset = pset  # keep the tables alive
        self.device = torch.device("cuda", pset.device)
        self.np = lib.pw_engine_npad(h)
        oh, ow, oc = c_int(), c_int(), c_int()
        check(lib.pw_engine_obs_shape(h, ctypes.byref(oh), ctypes.byref(ow), ctypes.byref(oc)))
        self.obs_shape = (oh.value, ow.value, oc.value)
        self.obs_bytes = lib.pw_engine_obs_bytes(h)
        self.obs_stride = lib.pw_engine_obs_stride(h)
        self.obs_dtype = torch.uint8 if obs_dtype == OBS_U8 else torch.float32
        for key, value in (options or {}).items():
            self.set_option(key, value)

    @property
    def render_kernel(self) -> str:
        """Name of the kernel ``pw_render`` launches for this engine (depends on the options)."""
        nb = ctypes.create_string_buffer(64)
        check(lib.pw_engine_render_kernel(self.handle, nb, 64))
        return nb.value.decode()

    def set_option(self, key, value) -> None:
        """``pw_engine_set_option``: ``key`` is a name of ``OPTIONS`` (or its number), ``value`` an integer or
        one of "group" / "wave" / "lane" (step_kernel), "auto" / "lds" (render_kernel)."""
        k = OPTIONS[key] if isinstance(key, str) else int(key)
        if isinstance(value, str):  # (a few names mean different numbers for different options)
            v = _OPTION_VALUES_BY_KEY.get(key, {}).get(value, _OPTION_VALUES.get(value))
            if v is None:
                raise ValueError(f"unknown value {value!r} for option {key!r}")
        else:
            v = int(value)
        check(lib.pw_engine_set_option(self.handle, k, v))

    def get_option(self, key) -> int:
        k = OPTIONS[key] if isinstance(key, str) else int(key)
        return check(lib.pw_engine_get_option(self.handle, k))

    def profile_render(self, launches: int) -> None:
        """Time the next ``launches`` render launches with HIP events on their own stream."""
        self.set_option("profile_render", launches)

    def profile_read(self):
        """Milliseconds of the render launches recorded since the last call (waits for them)."""
        cap = self.get_option("profile_render")
        buf = (ctypes.c_float * max(cap, 1))()
        n = check(lib.pw_engine_profile_read(self.handle, buf, cap))
        return [buf[i] for i in range(min(n, cap))]

    def tune_render(self, puzzle_id, pos, obs_storage) -> int:
        """``pw_engine_tune_render``: times the launch configurations of the page-ordered render kernel on these
        buffers, keeps the fastest (``page_order`` / ``page_run_log2`` / ``page_lds_pad_kb`` options) and returns
        its index; ``obs_storage`` holds the observations of ``pos`` afterwards."""
        return check(lib.pw_engine_tune_render(self.handle, _ptr(puzzle_id), _ptr(pos), _ptr(obs_storage),
                                               self.obs_stride, pos.shape[0], self._stream()))

    def bad_actions(self) -> int:
        """Out-of-range actions seen by this engine's step kernels since the last call (reads and clears the
        sticky device counter; synchronises the stream)."""
        return check(lib.pw_engine_bad_actions(self.handle, self._stream()))

    def validate(self, puzzle_id, pos=None) -> None:
        """``pw_validate_state``: raises ``ValueError`` when a puzzle id is outside the set or a movable outside
        its puzzle's grid (the kernels index with both unchecked).  Synchronises the stream."""
        first = c_int32(-1)
        n = check(lib.pw_validate_state(self.handle, _ptr(puzzle_id), _ptr(pos), puzzle_id.shape[0], ctypes.byref(first),
                                        self._stream()))
        if n:
            raise ValueError(f"{n} environment(s) with a puzzle id outside the set or a position outside the grid "
                             f"(first: environment {first.value})")

    def _stream(self):
        return c_void_p(_raw_stream(self.device.index))

    def counters(self):
        """``pw_counters``: dict of the device-side throughput counters (env_steps, episodes_ended, episodes_solved,
        bad_actions) summed over the engine's step launches so far.  Synchronises the stream."""
        out = (c_int64 * 4)()
        check(lib.pw_counters(self.handle, out, self._stream()))
        return {"env_steps": out[0], "episodes_ended": out[1], "episodes_solved": out[2], "bad_actions": out[3]}

    def counters_reset(self) -> None:
        check(lib.pw_counters_reset(self.handle, self._stream()))

    # pre-marshalled entry points: the state tensors of a vector environment never move, so their pointers are converted
    # once; a call then costs one ctypes call with ready arguments (the actions pointer and the stream are the only
    # per-call values).  The tensors are kept alive by the closure.
    def bind_step(self, puzzle_id, pos, steps, reward, dgoals, terminated, truncated, flags=0):
        """Returns ``call(actions_data_ptr)`` == ``step(puzzle_id, actions, pos, ...)`` on the current stream."""
        fn, h, dev, batch = lib.pw_step, self.handle, self.device.index, pos.shape[0]
        a = [_ptr(t) for t in (puzzle_id, pos, steps, reward, dgoals, terminated, truncated)]
        keep = (puzzle_id, pos, steps, reward, dgoals, terminated, truncated)

        def call(actions_ptr, _keep=keep):
            rc = fn(h, a[0], actions_ptr, a[1], a[2], a[3], a[4], a[5], a[6], batch, flags, _raw_stream(dev))
            if rc:
                check(rc)
        return call

    def bind_step_render(self, puzzle_id, pos, steps, reward, dgoals, terminated, truncated, obs_storage, flags=0,
                         delta=False):
        """Returns ``call(actions_data_ptr)`` == ``step_render(...)`` (``delta``: ``step_render_delta``)."""
        fn = lib.pw_step_render_delta if delta else lib.pw_step_render
        h, dev, batch, stride = self.handle, self.device.index, pos.shape[0], self.obs_stride
        a = [_ptr(t) for t in (puzzle_id, pos, steps, reward, dgoals, terminated, truncated, obs_storage)]
        keep = (puzzle_id, pos, steps, reward, dgoals, terminated, truncated, obs_storage)

        def call(actions_ptr, _keep=keep):
            rc = fn(h, a[0], actions_ptr, a[1], a[2], a[3], a[4], a[5], a[6], a[7], stride, batch, flags, _raw_stream(dev))
            if rc < 0:
                check(rc)
            return rc  # (1: the engine's completion word will be written -- pw_engine_set_step_signal)
        return call

    def set_step_signal(self, word) -> None:
        """``pw_engine_set_step_signal``: ``word`` = a pinned int64 / uint64 tensor of one element, or None."""
        check(lib.pw_engine_set_step_signal(self.handle, None if word is None else c_void_p(word.data_ptr())))
        self._step_signal_keep = word

    def set_step_host_copy(self, block) -> None:
        """``pw_engine_set_step_host_copy``: ``block`` = a pinned uint8 tensor of 16 + 2 * NP bytes, or None."""
        check(lib.pw_engine_set_step_host_copy(self.handle, None if block is None else c_void_p(block.data_ptr())))
        self._step_host_copy_keep = block

    def next_state(self, puzzle_index: int, xy_in, action: int, xy_out, info=None) -> None:
        """``pw_next_state``: host buffers in / out (bytes-like of 2 N int8 each), one launch, no copy command."""
        check(lib.pw_next_state(self.handle, puzzle_index, xy_in, action, xy_out, info))

    def plan_states(self, puzzle_index: int, actions: bytes, start=None, dev_states=None):
        """``pw_plan_states``: (states int8 [T + 1, N, 2], goal flags uint8 [T + 1]) as numpy arrays.

        A plan of any length (the reference's ``is_valid_plan`` / ``render_plan`` accept any, puzzle.py:413-424, 471-506):
        beyond ``PW_PLAN_MAX_ACTIONS`` actions per launch the plan continues from the last state of the previous chunk."""
        import numpy as np

        T = len(actions)
        n = self.pset.puzzles[puzzle_index].num_movables if self.pset.puzzles else None
        if n is None:
            raise ValueError("plan_states needs a puzzle set built from parsed puzzles")
        states = np.zeros((T + 1, n, 2), np.int8)
        goals = np.zeros((T + 1,), np.uint8)
        row = self.np * 2  # bytes per state of the device buffer (int8 [T + 1, np, 2])
        dev_base = None if dev_states is None else dev_states.data_ptr()
        t0 = 0
        while True:
            t1 = min(T, t0 + PLAN_MAX_ACTIONS)
            if t0 == 0:
                start_ptr = None if start is None else c_void_p(start.ctypes.data)
            else:  # the chunk starts where the previous one ended (its first state is rewritten with the same values)
                start_ptr = c_void_p(states[t0].ctypes.data)
            check(lib.pw_plan_states(self.handle, puzzle_index, start_ptr, actions[t0:t1], t1 - t0,
                                     c_void_p(states[t0:].ctypes.data), c_void_p(goals[t0:].ctypes.data),
                                     None if dev_base is None else c_void_p(dev_base + t0 * row)))
            if t1 >= T:
                break
            t0 = t1
        return states, goals

    # state buffers -------------------------------------------------------------------
    def alloc_state(self, batch: int):
        d = self.device
        return {
            "pos": torch.zeros((batch, self.np, 2), dtype=torch.int8, device=d),
            "steps": torch.zeros((batch,), dtype=torch.int32, device=d),
            "reward": torch.zeros((batch,), dtype=torch.float64, device=d),
            "dgoals": torch.zeros((batch,), dtype=torch.int8, device=d),
            "terminated": torch.zeros((batch,), dtype=torch.uint8, device=d),
            "truncated": torch.zeros((batch,), dtype=torch.uint8, device=d),
        }

    def alloc_obs(self, batch: int):
        """Returns (storage, view): ``view`` has shape (B, H, W, 3) over rows of
        ``obs_stride`` bytes (16-byte aligned environments)."""
        esz = 1 if self.obs_dtype == torch.uint8 else 4
        storage = torch.zeros((batch, self.obs_stride // esz), dtype=self.obs_dtype, device=self.device)
        h, w, c = self.obs_shape
        view = storage.as_strided((batch, h, w, c), (self.obs_stride // esz, w * c, c, 1))
        return storage, view

    def alloc_obs_host(self, batch: int):
        """``alloc_obs`` in pinned host memory (hipHostMalloc: mapped into the device's address space at the same address):
        the render kernels write it over PCIe, the host reads it without a copy command.  For the single-environment
        adapters -- a batch of thousands belongs in HBM."""
        esz = 1 if self.obs_dtype == torch.uint8 else 4
        storage = torch.zeros((batch, self.obs_stride // esz), dtype=self.obs_dtype).pin_memory()
        h, w, c = self.obs_shape
        view = storage.as_strided((batch, h, w, c), (self.obs_stride // esz, w * c, c, 1))
        return storage, view

    def _wrap_owned(self, ptr: int, batch: int):
        """(storage, view) tensors over a library-owned buffer; the memory goes back to the DEVICE (``pw_obs_free``)
        when the last tensor over it has gone."""
        esz = 1 if self.obs_dtype == torch.uint8 else 4
        holder = _OwnedObs(self, ptr, (batch, self.obs_stride // esz), "|u1" if esz == 1 else "<f4")
        try:
            storage = _dlpack_tensor(holder, self.device.index, esz)
        except Exception:  # noqa: BLE001 -- second route into torch: the CUDA array interface
            holder.owned_by_dlpack = False
            storage = torch.as_tensor(holder, device=self.device)
        if storage.data_ptr() != ptr or storage.dtype != self.obs_dtype or storage.device != self.device:
            raise RuntimeError("torch did not adopt the library-owned observation buffer in place")
        h, w, c = self.obs_shape
        view = storage.as_strided((batch, h, w, c), (self.obs_stride // esz, w * c, c, 1))
        return storage, view

    def alloc_obs_owned(self, batch: int):
        """``pw_obs_alloc``: like ``alloc_obs`` but the buffer is owned by the library (HIP virtual-memory API), not
        by torch's caching allocator."""
        p = c_void_p()
        check(lib.pw_obs_alloc(self.handle, batch, ctypes.byref(p)))
        return self._wrap_owned(p.value, batch)

    def alloc_obs_tuned(self, puzzle_id, pos, max_candidates: int):
        """``pw_obs_alloc_tuned``: returns (storage, view, tuned index, [ms of every candidate tried]); the buffer
        holds the observations of ``pos``, the engine the tuned launch configuration."""
        p, tried = c_void_p(), c_int32(0)
        k = max(1, int(max_candidates))
        ms = (ctypes.c_float * k)()
        idx = check(lib.pw_obs_alloc_tuned(self.handle, _ptr(puzzle_id), _ptr(pos), pos.shape[0], k, ctypes.byref(p), ms,
                                           ctypes.byref(tried), self._stream()))
        storage, view = self._wrap_owned(p.value, pos.shape[0])
        return storage, view, idx, [float(ms[i]) for i in range(tried.value)]

    # kernels ---------------------------------------------------------------------------
    def reset(self, puzzle_id, pos, steps, terminated=None, truncated=None, mask=None):
        check(lib.pw_reset(self.handle, _ptr(puzzle_id), _ptr(mask), _ptr(pos), _ptr(steps), _ptr(terminated),
                           _ptr(truncated), pos.shape[0], self._stream()))

    def step(self, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, flags=0):
        check(lib.pw_step(self.handle, _ptr(puzzle_id), _ptr(actions), _ptr(pos), _ptr(steps), _ptr(reward),
                          _ptr(dgoals), _ptr(terminated), _ptr(truncated), pos.shape[0], flags, self._stream()))

    def resample(self, puzzle_id, episode, seed, terminated=None, truncated=None, table=None):
        """Finished environments (flags set; both None = all) draw their next puzzle (``pw_resample``)."""
        if table is not None and table.numel() == 0:
            raise ValueError("pw_resample: empty sampling table")
        check(lib.pw_resample(self.handle, _ptr(puzzle_id), _ptr(terminated), _ptr(truncated), _ptr(table),
                              0 if table is None else table.shape[0], int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(episode),
                              puzzle_id.shape[0], self._stream()))

    def rollout(self, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, reward_hist=None,
                terminated_hist=None, truncated_hist=None, flags=0):
        """``actions``: uint8 [T, B] (step-major)."""
        check(lib.pw_rollout(self.handle, _ptr(puzzle_id), _ptr(actions), actions.shape[0], _ptr(pos), _ptr(steps),
                             _ptr(reward), _ptr(dgoals), _ptr(terminated), _ptr(truncated), _ptr(reward_hist),
                             _ptr(terminated_hist), _ptr(truncated_hist), pos.shape[0], flags, self._stream()))

    def bind(self, puzzle_id) -> dict:
        """``pw_batch_bind``: from now on ``step`` / ``rollout`` / ``step_render`` calls that pass THIS ``puzzle_id`` tensor run
        one lane per environment with the puzzle's push tables in LDS, for every puzzle that at least ``bind_min_envs``
        environments of the batch play (the others keep the lane groups, in the same launch).  The ids may only change through
        ``resample`` / before a ``reset`` on this tensor (both rebuild the binding), else bind again."""
        info = (c_int64 * 6)()
        check(lib.pw_batch_bind(self.handle, _ptr(puzzle_id), puzzle_id.shape[0], info, self._stream()))
        self._bound = puzzle_id  # (the engine keeps the pointer: the tensor must stay alive)
        # (lane_envs: environments no segment holds that launches of several steps step one lane each, 64 puzzles per wavefront)
        return {"segments": info[0], "bound_envs": info[1], "bound_puzzles": info[2], "listed_puzzles": info[3], "lane_envs": info[4]}

    def unbind(self) -> None:
        check(lib.pw_batch_unbind(self.handle))
        self._bound = None

    def mailbox(self, puzzle_id, pos, steps, reward, dgoals, terminated, truncated, flags=0, ring=8, idle_ms=1000):
        """``pw_mailbox_open``: the resident step kernel over these arrays (see :class:`Mailbox`)."""
        return Mailbox(self, puzzle_id, pos, steps, reward, dgoals, terminated, truncated, flags, ring, idle_ms)

    def render(self, puzzle_id, pos, obs_storage):
        check(lib.pw_render(self.handle, _ptr(puzzle_id), _ptr(pos), _ptr(obs_storage), self.obs_stride,
                            pos.shape[0], self._stream()))

    def step_render(self, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, obs_storage,
                    flags=0):
        check(lib.pw_step_render(self.handle, _ptr(puzzle_id), _ptr(actions), _ptr(pos), _ptr(steps),
                                 _ptr(reward), _ptr(dgoals), _ptr(terminated), _ptr(truncated),
                                 _ptr(obs_storage), self.obs_stride, pos.shape[0], flags, self._stream()))

    def step_render_delta(self, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, obs_storage,
                          flags=0):
        """``pw_step_render_delta``: ``obs_storage`` must hold the observation of ``pos`` on entry."""
        check(lib.pw_step_render_delta(self.handle, _ptr(puzzle_id), _ptr(actions), _ptr(pos), _ptr(steps),
                                       _ptr(reward), _ptr(dgoals), _ptr(terminated), _ptr(truncated),
                                       _ptr(obs_storage), self.obs_stride, pos.shape[0], flags, self._stream()))

    def expand4(self, puzzle_index, states, succ, moved, goal):
        check(lib.pw_expand4(self.handle, int(puzzle_index), _ptr(states), _ptr(succ), _ptr(moved), _ptr(goal),
                             states.shape[0], self._stream()))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and lib is not None:
            mb = getattr(self, "_mailbox", None)
            mb = mb() if mb is not None else None
            if mb is not None:
                mb.handle = None  # (pw_engine_destroy closes an open mailbox itself)
            lib.pw_engine_destroy(h)
            self.handle = None
