"""Single-environment core shared by the gym and dm_env adapters.

Reproduces the constructor / reset / step bookkeeping of
python3/src/pushworld/gym_env.py:57-226 and dm_env.py:60-234 around a batch-1 engine.  The
puzzle pool is ONE packed puzzle set in HBM; ``reset`` only changes the environment's
puzzle id.
"""
from __future__ import annotations

import os
import random
from typing import Optional

import numpy as np
import torch

from . import _capi
from .config import PUZZLE_EXTENSION
from .puzzle import PushWorldPuzzle, default_device_index
from .utils.env_utils import get_max_puzzle_dimensions
from .utils.filesystem import iter_files_with_extension


def pool_frame(puzzles, standard_padding: bool):
    """(height, width) in cells of the observation frame of a puzzle pool (gym_env.py:78-103): the pool maximum,
    or the benchmark maximum with ``standard_padding`` (ValueError when the pool does not fit into it)."""
    widths, heights = zip(*[p.dimensions for p in puzzles])
    max_w, max_h = max(widths), max(heights)
    if standard_padding:
        std_h, std_w = get_max_puzzle_dimensions()
        for what, std, own in (("height", std_h, max_h), ("width", std_w, max_w)):
            if std < own:
                raise ValueError(
                    f"`standard_padding` is True, but the maximum puzzle {what} in BENCHMARK_PUZZLES_PATH is "
                    f"less than the {what} of the puzzle(s) in the given `puzzle_path`."
                )
        max_h, max_w = std_h, std_w
    return max_h, max_w


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

        self._max_cell_height, self._max_cell_width = pool_frame(self._puzzles, standard_padding)

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
        # all per-step scalars and the positions live in ONE small buffer (typed views into it) in pinned HOST memory, which the
        # device addresses too: the step kernel reads and writes it over PCIe, and the host reads it after the one stream
        # synchronisation of a step -- no copy command (a device buffer + an 80-byte hipMemcpyAsync before: the copy engine's
        # latency was a third of the step, 18.5 k -> 22.8 k gym steps/s on a Level-1 puzzle; tools/experiments/c1_hostobs_xp.py)
        dev_t = self._engine.device
        npad = self._engine.np
        self._raw = torch.zeros((16 + 2 * npad,), dtype=torch.uint8).pin_memory()
        # Where the step is ONE launch (frames of at least 64 KiB: the gym default), the state itself lives in DEVICE memory and that launch
        # writes a copy of what it left into self._raw (pw_engine_set_step_host_copy): a kernel that reads and writes pinned host
        # memory across the link lasts ~5 us whatever it computes (profiles/r06_c1_trace.txt).  PUSHWORLD_AMD_DEVICE_STATE=0: as before.
        self._device_state = (self._engine.get_option("step_one_applies") == 1
                              and os.environ.get("PUSHWORLD_AMD_DEVICE_STATE", "1") != "0")
        self._dev_raw = torch.zeros((16 + 2 * npad,), dtype=torch.uint8, device=dev_t) if self._device_state else None
        src = self._dev_raw if self._device_state else self._raw
        self._buf = {
            "reward": src[0:8].view(torch.float64),
            "steps": src[8:12].view(torch.int32),
            "terminated": src[12:13],
            "truncated": src[13:14],
            "dgoals": src[14:15].view(torch.int8),
            "pos": src[16:].view(torch.int8).view(1, npad, 2),
        }
        self._pid = torch.zeros((1,), dtype=torch.int32, device=dev_t)
        self._acts = torch.arange(4, dtype=torch.uint8, device=dev_t)  # action a = the 1-element view [a : a + 1]
        # The observation lives in pinned HOST memory, which the device addresses too: the render kernels write it over
        # PCIe -- a step only the pixel rows its moved objects swept (pw_step_render_delta) -- and no copy command runs
        # (9.7 k -> 18 k gym steps/s on the C1 puzzle at max_steps 50; a device buffer + hipMemcpy of the whole frame before)
        self._obs_storage, self._obs = self._engine.alloc_obs_host(1)
        self.obs_shape = self._engine.obs_shape
        self._raw_np = self._raw.numpy()
        self._obs_np = self._obs[0].numpy()  # (a view: made once, copied per step)
        # the step's pointers never change: marshalled once (the action is a byte of self._acts)
        b = self._buf
        self._step_call = self._engine.bind_step_render(self._pid, b["pos"], b["steps"], b["reward"], b["dgoals"], b["terminated"],
                                                        b["truncated"], self._obs_storage, 0, delta=True)
        self._acts_ptr = self._acts.data_ptr()
        # ... and the last kernel of a step says when everything is written: the host polls this word instead of synchronising
        # the stream (the runtime's synchronisation costs ~8 us a step)
        self._signal = torch.zeros((1,), dtype=torch.int64).pin_memory()
        self._signal_np = self._signal.numpy()
        self._signalled = 0
        self._engine.set_step_signal(self._signal)
        if self._device_state:
            self._engine.set_step_host_copy(self._raw)
        self._stream = torch.cuda.current_stream(dev_t)
        # One GRAPH launch per step: the step's two launches (step kernel, redraw of the changed rows + completion
        # word) are captured once per action -- the action is a byte of self._acts, i.e. part of the captured pointer; four graphs --
        # and replayed: one runtime call instead of two kernel launches.  Captured after a few eager steps (first calls may
        # allocate); PUSHWORLD_AMD_STEP_GRAPHS=0 keeps the eager launches (A/B runs).
        self._graphs = None if os.environ.get("PUSHWORLD_AMD_STEP_GRAPHS", "1") != "0" else False
        self._eager_steps = 0

    # ------------------------------------------------------------------
    def _read_back(self, signalled: bool = False):
        """The observation, the scalars and the positions are in pinned host memory already: wait for the step's completion word
        (or, without one, synchronise the stream)."""
        if signalled:
            word, want = self._signal_np, self._signalled
            for _ in range(200000):  # (~20 ms: then ask the runtime -- a failed launch must not spin for ever)
                if word[0] >= want:  # (>=: the engine's count can only run ahead of ours -- see the re-arm below)
                    break
            else:
                # The word did not arrive: our count and the engine's have come apart (an exception between the launch and the
                # count's increment, another caller's pw_step_render_delta on this engine) or the launch failed.  The stream
                # says when the step is done; the word is armed again, both counts from zero.
                torch.cuda.current_stream(self._engine.device).synchronize()
                self._signal_np[0] = 0
                self._signalled = 0
                self._engine.set_step_signal(self._signal)
            # (the observation / scalars below were written before the word by the same kernel with system-scope stores; this
            # host's loads are not reordered before the load that saw the word -- x86; another host ISA needs an acquire fence here)
        else:
            torch.cuda.current_stream(self._engine.device).synchronize()
        return self._obs_np.copy(), self._raw_np

    def _capture_graphs(self) -> None:
        """Four HIP graphs, one per action, of this environment's step (``torch.cuda.CUDAGraph`` over the library's launches: they go
        to torch's current stream and neither allocate nor synchronise).  A captured launch does not run, but the engine counted
        it: the completion word is armed again afterwards; a replayed step writes the (non-zero) number its capture baked in, so
        the host zeroes the word before a replay and waits for anything else."""
        dev = self._engine.device
        try:
            torch.cuda.synchronize(dev)
            graphs = []
            for a in range(4):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    rc = self._step_call(self._acts_ptr + a)
                if rc != 1:
                    raise RuntimeError("the captured step does not write the completion word")
                graphs.append(g)
            torch.cuda.synchronize(dev)
            self._graphs = graphs
        except Exception:  # noqa: BLE001  (a runtime without graph capture: the eager launches stay)
            self._graphs = False
            torch.cuda.synchronize(dev)
        self._signal_np[0] = 0
        self._signalled = 0
        self._engine.set_step_signal(self._signal)

    def _graph_step(self, action: int):
        word = self._signal_np
        word[0] = 0
        self._graphs[action].replay()
        for _ in range(200000):
            if word[0] != 0:
                break
        else:
            torch.cuda.current_stream(self._engine.device).synchronize()
        return self._obs_np.copy(), self._raw_np

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
        n = self._current_puzzle.num_movables
        self._xy_view = self._raw_np[16:16 + 2 * n].view(np.int8)  # (x, y) of the puzzle's movables as the step kernel leaves them
        self._reward_view = self._raw_np[0:8].view(np.float64)
        self._flags_view = self._raw_np[12:14]                     # terminated, truncated
        return self._read_back()[0]

    def core_step(self, action: int):
        """Returns (observation, reward: float, terminated: bool, truncated: bool)."""
        if self._current_state is None:
            raise RuntimeError("reset() must be called before step() can be called.")
        action = int(action)
        if not 0 <= action <= 3:  # (the adapters have checked their action spaces; the pointer arithmetic below must not run wild)
            raise ValueError("The provided action is not in the action space.")
        # the observation buffer is this environment's own and always current: incremental redraw (pw_step_render_delta)
        if self._graphs:
            observation, raw = self._graph_step(action)
        else:
            rc = self._step_call(self._acts_ptr + action)
            signalled = rc >= 1
            if signalled:
                self._signalled += 1
            observation, raw = self._read_back(signalled)
            self._eager_steps += 1
            if rc == 2:
                self._graphs = False  # step + redraw are ONE launch already (PW_OPT_STEP_ONE_FUSED): cheaper than a graph replay, and not replayable
            elif self._device_state:
                # (the one launch did not run -- its option was switched off behind this object's back: the state is on the device, fetch it)
                self._raw.copy_(self._dev_raw)
                torch.cuda.current_stream(self._engine.device).synchronize()
                self._graphs = False
            if self._graphs is None and signalled and self._eager_steps >= 8:
                self._capture_graphs()
        self._steps += 1
        # (views made once per reset, one tolist() each: 3.4 -> 0.7 us of Python per step)
        it = iter(self._xy_view.tolist())
        self._current_state = tuple(zip(it, it))
        flags = self._flags_view.tolist()
        return observation, self._reward_view.item(), flags[0] != 0, flags[1] != 0

    def core_render_u8(self) -> np.ndarray:
        """uint8, unpadded: puzzle.render(current_state) (gym_env.py:228-240)."""
        return self._current_puzzle.render(self._current_state, border_width=self._border_width,
                                           pixels_per_cell=self._pixels_per_cell)

