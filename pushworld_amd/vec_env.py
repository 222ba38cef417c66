"""Batched PushWorld environments on one MI355X.

``VecPushWorld`` is the batched form of the reference's ``PushWorldEnv.step/reset``
(python3/src/pushworld/gym_env.py:150-226): B independent environments, each bound to one
puzzle of a pool, stepped by one kernel launch (+ one render launch).  All state lives in
HBM as torch tensors; nothing is copied to the host per step.

Semantics per environment are exactly the reference's: no implicit reset after
termination (trap T9) unless ``autoreset=True`` (next-step autoreset, gymnasium's vector
convention: the call after a finished episode resets that environment and returns reward 0).
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import _capi
from .puzzle import DEFAULT_BORDER_WIDTH, DEFAULT_PIXELS_PER_CELL, PushWorldPuzzle, default_device_index


class VecPushWorld:
    """B environments over a pool of puzzles.

    Args:
        puzzles: sequence of ``PushWorldPuzzle`` objects or ``.pwp`` paths (the pool), or a
            ``_capi.PuzzleSet`` (e.g. ``PuzzleSet.load(path, device)`` of a packed pool file).
        num_envs: B.
        puzzle_ids: int array [B] of pool indices (default: ``i % len(puzzles)``).  Grouping
            equal ids together keeps a workgroup's puzzle tables hot in L1/L2.
        max_steps: truncation limit (gym_env.py:223) or None.
        observation: "uint8", "float32" or None (state only, no render kernel).
        pad_cells: (height, width) of the observation frame in cells; default = pool maximum
            (gym_env.py:80-82).
        autoreset: next-step autoreset inside the step kernel.
        fused: one ``pw_step_render`` call per step (default: the step kernel hands its page records to the
            page-ordered render, no pre-pass) instead of ``pw_step`` + ``pw_render``.
        incremental: keep the observation buffer up to date with ``pw_step_render_delta``: a step
            rewrites only the pixel rows swept by the objects that moved (the buffer persists between
            steps, so everything else is already right).  Same observations, a fraction of the HBM
            writes; uint8 / pixels_per_cell 3 only (other settings fall back to the full render).
        resample: draw a new puzzle for every new episode ON THE DEVICE (``pw_resample``), the batched
            form of ``random.choice(self._puzzles)`` in gym_env.py:172.  ``True`` = uniform over the
            pool; a sequence of pool indices = sampling table (repeat an index to weight it).
            Draws happen in ``reset`` (all / masked environments) and, with ``autoreset``, at the
            start of the ``step`` that resets a finished environment.
        seed: seed of the counter-based draw: puzzle = f(seed, environment index, episode number).
        tune: auto-tune the launch configuration of the page-ordered render kernel on this environment's own
            observation buffer (``pw_engine_tune_render``, a few dozen extra render launches in the constructor).
            Default: on for observation buffers of 256 MB and more, where the choice is worth 5-15 %.
        tune_allocations: with ``tune``: the observation buffer is allocated BY THE LIBRARY (``pw_obs_alloc_tuned``:
            physical chunks mapped with the HIP virtual-memory API) from up to this many candidate allocations
            (default: up to 32, within a third of the device memory: 25 for the C3 batch).  What the HBM-write-bound render reaches
            on a buffer follows the buffer's physical backing -- two classes, 7-10 % apart, per allocation
             -- so the library takes a quick look (~10 ms) at one candidate after the other, all
            alive at once, keeps the first of the fast class, releases the others to the DEVICE and tunes on the kept one (nothing stays behind in
            torch's caching allocator).  ``self.obs`` is bound once, in the constructor.  0: a plain torch buffer,
            tuned in place.
        bind: ``pw_batch_bind`` the puzzle assignment at every ``reset`` (and behind every device-side ``resample``): puzzles played by
            at least 48 environments of the batch are stepped one lane per environment with their push tables in LDS instead of by
            lane groups walking the tables through the caches.  Default (None): on, except for ``incremental`` and ``resample`` (a
            binding is rebuilt behind every ``pw_resample``); a set of 8 x 8 puzzles keeps its whole-grid boards unless the segments
            hold every environment.  Same results either way.
        engine_options: ``pw_engine_set_option`` settings (``_capi.OPTIONS``), e.g. ``{"step_kernel": "lane"}`` --
            kernel selection for tests and A/B runs; results never depend on them.
    """

    def __init__(self, puzzles: Sequence[Union[str, PushWorldPuzzle]], num_envs: int,
                 puzzle_ids: Optional[Sequence[int]] = None, max_steps: Optional[int] = None,
                 border_width: int = DEFAULT_BORDER_WIDTH, pixels_per_cell: int = DEFAULT_PIXELS_PER_CELL,
                 observation: Optional[str] = "float32", pad_cells=None, device: Optional[int] = None,
                 autoreset: bool = False, fused: bool = True, resample=False, seed: int = 0,
                 incremental: bool = False, engine_options: Optional[dict] = None, tune: Optional[bool] = None,
                 tune_allocations: Optional[int] = None, bind: Optional[bool] = None):
        if observation not in ("uint8", "float32", None):
            raise ValueError("observation must be 'uint8', 'float32' or None")
        dev = default_device_index() if device is None else int(device)
        if isinstance(puzzles, _capi.PuzzleSet):  # e.g. PuzzleSet.load(packed file): no texts, no parsing
            if puzzles.device != dev:
                raise ValueError(f"the puzzle set lives on device {puzzles.device}, not {dev}")
            self.pset = puzzles
            self.puzzles = None
        else:
            self.puzzles = [p if isinstance(p, PushWorldPuzzle) else PushWorldPuzzle(p) for p in puzzles]
            if not self.puzzles:
                raise ValueError("No PushWorld puzzles given")
            self.pset = _capi.PuzzleSet([p._parsed for p in self.puzzles], dev)
        self.num_puzzles = len(self.pset)
        ph, pw = pad_cells if pad_cells is not None else (0, 0)
        dtype = _capi.OBS_F32 if observation == "float32" else _capi.OBS_U8
        # (max_batch: the engine's per-environment scratch is sized here, once -- no step / render call allocates later)
        self.engine = _capi.Engine(self.pset, max_steps, pixels_per_cell, border_width, dtype, ph, pw,
                                   options=engine_options, max_batch=int(num_envs) if observation is not None else 0)
        self.device = self.engine.device
        self.num_envs = int(num_envs)
        self.observation = observation
        self.flags = _capi.STEP_AUTORESET if autoreset else 0
        self.fused = bool(fused)
        self.incremental = bool(incremental)
        self._obs_current = False  # the observation buffer holds the observation of self.pos
        if bind is None:
            # (not with `resample`: every step would rebuild the binding behind its pw_resample -- three small launches)
            # (sets of 8 x 8 puzzles too: the engine takes the segments where they hold EVERY environment -- C2 -- and its whole-grid
            # boards otherwise)
            bind = not self.incremental and self.engine.get_option("bind_puzzles") > 0 \
                and "step_kernel" not in (engine_options or {}) and (resample is False or resample is None)
        self._bind = bool(bind)
        self.bound_info = None  # what the last pw_batch_bind reported (segments, bound environments / puzzles)

        if puzzle_ids is None:
            ids = np.arange(self.num_envs) % self.num_puzzles
        else:
            ids = np.asarray(puzzle_ids)
            if ids.shape != (self.num_envs,) or ids.min() < 0 or ids.max() >= self.num_puzzles:
                raise ValueError("puzzle_ids must be [num_envs] indices into the puzzle pool")
        self.puzzle_id = torch.as_tensor(ids, dtype=torch.int32).to(self.device)
        if "step_block_order" not in (engine_options or {}) and self.num_envs >= 4096 and not resample:
            # A launch lasts as long as its slowest wavefronts: when the expensive puzzles (many movables) sit at the
            # END of the batch (pools sorted by level), let the step kernel start there and the cheap ones fill the tail.
            # (Only for a fixed assignment: with `resample` the puzzles of the batch change every episode.)
            hdr = np.frombuffer(self.pset.headers(), np.uint8)
            assert hdr.size == _capi.PUZZLE_HEADER_BYTES * self.num_puzzles, "PwPuzzleHeader layout changed (csrc/pw_format.h)"
            n_mov = hdr.reshape(-1, _capi.PUZZLE_HEADER_BYTES)[:, _capi.PUZZLE_HEADER_N_OFFSET].astype(np.float64)
            cost = n_mov[np.asarray(ids, dtype=np.int64)]
            q = max(1, self.num_envs // 4)
            if cost[-q:].mean() > 1.15 * cost[:q].mean():
                self.engine.set_option("step_block_order", "reverse")
        st = self.engine.alloc_state(self.num_envs)
        self.pos, self.steps = st["pos"], st["steps"]
        self.reward, self.dgoals = st["reward"], st["dgoals"]
        self.terminated, self.truncated = st["terminated"], st["truncated"]
        self.tuned_config = None  # index returned by the tuner, once it ran
        self.tuned_ms = None      # milliseconds per render launch it measured for that configuration
        self.tuned_candidates_ms = []  # the same for every candidate allocation it tried (tune_allocations)
        self._has_reset = False
        self._obs_storage, self.obs = None, None
        self.obs_owned_by_library = False
        if observation is not None:
            nbytes = self.num_envs * self.engine.obs_stride
            if tune is None:
                tune = nbytes >= (256 << 20)
            if tune and tune_allocations is None:
                # At most 32 candidates, all alive at once while the choice is made: within a third of the device memory
                # AND within what is free right now (a co-resident model or torch's own allocator must not be pushed into
                # OOM during the constructor); torch.cuda.mem_get_info() was seen returning 0 free bytes on these boxes:
                # then a small count, and the library stops at the candidates there are when memory runs out.
                # (On a box whose allocations are mostly of the slow class -- one in six to thirteen fast,
                # profiles/r03_bench_final7.json: the 14th candidate was the first fast one -- twelve candidates miss one
                # time in three; a candidate costs ~40 ms and goes back to the device.)
                total = torch.cuda.get_device_properties(self.device).total_memory
                try:
                    free = int(torch.cuda.mem_get_info(self.device)[0])
                except Exception:  # noqa: BLE001
                    free = 0
                budget = min(total // 3, free // 2) if free > 0 else 4 * nbytes
                tune_allocations = min(32, max(1, int(budget // nbytes)))
            owned = False
            if tune and tune_allocations:
                # library-owned buffer: candidates are tuned on the initial states (reset() draws them again)
                self.engine.reset(self.puzzle_id, self.pos, self.steps, self.terminated, self.truncated, None)
                try:
                    self._obs_storage, self.obs, self.tuned_config, self.tuned_candidates_ms = \
                        self.engine.alloc_obs_tuned(self.puzzle_id, self.pos, int(tune_allocations))
                    self.tuned_ms = self.engine.get_option("tuned_ns") * 1e-6
                    owned = True
                except (RuntimeError, MemoryError) as exc:
                    # the HIP virtual-memory API refused (old runtime, exotic device): the same kernels on a torch buffer
                    import warnings

                    warnings.warn(f"pw_obs_alloc_tuned failed ({exc}); using a torch-owned observation buffer", RuntimeWarning)
            self.obs_owned_by_library = owned
            if not owned:
                self._obs_storage, self.obs = self.engine.alloc_obs(self.num_envs)
                if tune:
                    self.engine.reset(self.puzzle_id, self.pos, self.steps, self.terminated, self.truncated, None)
                    self.tuned_config = self.engine.tune_render(self.puzzle_id, self.pos, self._obs_storage)
                    self.tuned_ms = self.engine.get_option("tuned_ns") * 1e-6
                    self.tuned_candidates_ms = [self.tuned_ms]

        self.seed = int(seed)
        self.resample = resample is not False and resample is not None
        self.sample_table = None
        if self.resample and resample is not True:
            tab = np.asarray(resample)
            if tab.ndim != 1 or tab.size == 0 or tab.min() < 0 or tab.max() >= self.num_puzzles:
                raise ValueError("resample must be True or a non-empty 1-D sequence of puzzle pool indices")
            self.sample_table = torch.as_tensor(tab, dtype=torch.int32).to(self.device)
        # episode number of every environment (uint32 counter bits in an int32 tensor)
        self.episode = torch.zeros((self.num_envs,), dtype=torch.int32, device=self.device)
        # pre-marshalled step entry points (the state tensors never move: their pointers are converted once)
        self._act_shape = (self.num_envs,)
        self._call_step = self.engine.bind_step(self.puzzle_id, self.pos, self.steps, self.reward, self.dgoals,
                                                self.terminated, self.truncated, self.flags)
        self._call_step_render = self._call_step_delta = None
        if self.obs is not None:
            self._call_step_render = self.engine.bind_step_render(
                self.puzzle_id, self.pos, self.steps, self.reward, self.dgoals, self.terminated, self.truncated,
                self._obs_storage, self.flags)
            self._call_step_delta = self.engine.bind_step_render(
                self.puzzle_id, self.pos, self.steps, self.reward, self.dgoals, self.terminated, self.truncated,
                self._obs_storage, self.flags, delta=True)

    # --------------------------------------------------------------------------------
    @property
    def num_objects_padded(self) -> int:
        return self.engine.np

    def set_puzzle_ids(self, puzzle_ids) -> None:
        self._obs_current = False
        self.puzzle_id.copy_(torch.as_tensor(np.asarray(puzzle_ids), dtype=torch.int32))
        if self._bind and self.bound_info is not None:
            self.bound_info = self.engine.bind(self.puzzle_id)

    def reset(self, mask: Optional[torch.Tensor] = None, seed: Optional[int] = None):
        """gym_env.py:150-186 for every (masked) environment; returns the observation tensor.
        ``seed`` (with ``resample``) restarts the puzzle draw sequence: episode numbers return to 0."""
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8)
        if seed is not None:
            self.seed = int(seed)
            self.episode.zero_()
        if self.resample:
            self.engine.resample(self.puzzle_id, self.episode, self.seed, terminated=mask, table=self.sample_table)
        self.engine.reset(self.puzzle_id, self.pos, self.steps, self.terminated, self.truncated, mask)
        self._has_reset = True
        if self._bind and (self.bound_info is None or self.resample or mask is None):
            self.bound_info = self.engine.bind(self.puzzle_id)  # (reads the segment count back: later launches have the exact grid)
        if self.obs is not None:
            self.engine.render(self.puzzle_id, self.pos, self._obs_storage)
            self._obs_current = True
        return self.obs

    def step(self, actions: torch.Tensor):
        """gym_env.py:188-226 for every environment.

        Args:
            actions: uint8 tensor [B] on the device with values in 0..3.

        Returns ``(obs, reward float64[B], terminated uint8[B], truncated uint8[B])`` --
        views of the engine's persistent HBM buffers (overwritten by the next call).
        """
        if not self._has_reset:
            raise RuntimeError("reset() must be called before step() can be called.")
        if actions.dtype != torch.uint8 or actions.device != self.device or actions.shape != self._act_shape \
                or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous uint8 tensor of shape [num_envs] on the engine's device")
        if self.resample and self.flags & _capi.STEP_AUTORESET:
            # finished environments draw the puzzle their autoreset (inside this step) starts from
            self.engine.resample(self.puzzle_id, self.episode, self.seed, self.terminated, self.truncated,
                                 self.sample_table)
        if self.obs is None:
            self._call_step(actions.data_ptr())  # one ctypes call with ready arguments: the launch is host bound
            return None, self.reward, self.terminated, self.truncated
        if self.incremental and self._obs_current:
            self._call_step_delta(actions.data_ptr())
        elif self.fused:
            self._call_step_render(actions.data_ptr())
        else:
            self._call_step(actions.data_ptr())
            self.engine.render(self.puzzle_id, self.pos, self._obs_storage)
        self._obs_current = True
        return self.obs, self.reward, self.terminated, self.truncated

    def counters(self) -> dict:
        """Device-side throughput counters of this environment's engine (``pw_counters``): env_steps, episodes_ended,
        episodes_solved, bad_actions since construction (or ``counters_reset``).  Synchronises the stream."""
        return self.engine.counters()

    def counters_reset(self) -> None:
        self.engine.counters_reset()

    def rollout(self, actions: torch.Tensor, history: bool = False):
        """T steps in ONE launch without observations (``pw_rollout``): ``actions`` is a uint8 tensor
        [T, B] on the device.  Same semantics as T calls of ``step`` with ``observation=None``.

        Returns ``(reward, terminated, truncated)`` of the last step, or, with ``history=True``,
        the per-step tensors of shape [T, B].
        """
        if not self._has_reset:
            raise RuntimeError("reset() must be called before step() can be called.")
        if actions.dtype != torch.uint8 or actions.device != self.device or actions.dim() != 2 \
                or actions.shape[1] != self.num_envs or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous uint8 tensor of shape [T, num_envs] on the engine's device")
        T = actions.shape[0]
        rh = th = uh = None
        if history:
            rh = torch.empty((T, self.num_envs), dtype=torch.float64, device=self.device)
            th = torch.empty((T, self.num_envs), dtype=torch.uint8, device=self.device)
            uh = torch.empty((T, self.num_envs), dtype=torch.uint8, device=self.device)
        self._obs_current = False
        self.engine.rollout(self.puzzle_id, actions, self.pos, self.steps, self.reward, self.dgoals, self.terminated,
                            self.truncated, rh, th, uh, self.flags)
        if history:
            return rh, th, uh
        return self.reward, self.terminated, self.truncated

    def mailbox(self, ring: int = 8, idle_ms: int = 1000):
        """Resident stepping (``pw_mailbox_open``) for hosts that need every step's verdicts before they choose the next actions:
        ``with vec.mailbox() as mb: reward, terminated, truncated = mb.step(actions)`` -- numpy views of pinned host memory, a
        few microseconds per step instead of a launch and a stream synchronisation.  Same semantics as ``step`` with
        ``observation=None`` (this object's ``pos`` / ``steps`` / ``reward`` / ... tensors are kept up to date); state only, up to
        65 536 environments; ``step`` / ``rollout`` / ``reset`` of this object fail while it is open.  Bad
        actions are flagged 0xFF in ``terminated`` / ``truncated`` and counted (``counters()['bad_actions']``) as in ``step``."""
        if not self._has_reset:
            raise RuntimeError("reset() must be called before step() can be called.")
        self._obs_current = False
        return self.engine.mailbox(self.puzzle_id, self.pos, self.steps, self.reward, self.dgoals, self.terminated, self.truncated,
                                   self.flags, ring, idle_ms)

    def render(self):
        """Re-renders the current states into the observation buffer and returns it."""
        if self.obs is None:
            raise RuntimeError("this VecPushWorld was created with observation=None")
        self.engine.render(self.puzzle_id, self.pos, self._obs_storage)
        self._obs_current = True
        return self.obs

    def states(self) -> np.ndarray:
        """Host copy of the positions, int8 [B, NP, 2] (entries past a puzzle's movables are 0)."""
        return self.pos.cpu().numpy()

    def set_states(self, pos: np.ndarray) -> None:
        self.pos.copy_(torch.as_tensor(np.asarray(pos), dtype=torch.int8))
        self._has_reset = True
        self._obs_current = False
