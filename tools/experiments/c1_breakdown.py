#!/usr/bin/env python3
"""Round-4 probe: where a gym step of the single-environment adapter spends its time (host timers around its parts)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pushworld_amd.gym_env import PushWorldEnv
from pushworld_amd.config import BENCHMARK_PUZZLES_PATH
import glob

path = sorted(glob.glob(os.path.join(BENCHMARK_PUZZLES_PATH, "level1", "*.pwp")))[0]
env = PushWorldEnv(path, max_steps=1000)
env.reset(seed=0)
core = env
b = core._buf
eng = core._engine
rng = np.random.default_rng(0)
acts = rng.integers(0, 4, 4000)
T = {k: 0.0 for k in ("launch", "copy", "sync", "npcopy", "parse", "whole")}
pc = time.perf_counter
for a in acts[:200]:
    env.step(int(a))
n = 0
for a in acts:
    t0 = pc()
    signalled = core._step_call(core._acts_ptr + int(a)) == 1
    t1 = pc()
    t2 = t1  # (round 5: no copy command -- the scalars live in pinned host memory)
    if signalled:  # (round 5: the step's completion word instead of a stream synchronisation)
        core._signalled += 1
        word, want = core._signal_np, core._signalled
        while word[0] != want:
            pass
    else:
        torch.cuda.current_stream(eng.device).synchronize()
    t3 = pc()
    o = core._obs[0].numpy().copy()
    t4 = pc()
    raw = core._raw_np
    xy = raw[16:16 + 6].view(np.int8)
    st = tuple((int(xy[2 * j]), int(xy[2 * j + 1])) for j in range(3))
    r = float(raw[0:8].view(np.float64)[0]); te = bool(raw[12]); tr = bool(raw[13])
    t5 = pc()
    T["launch"] += t1 - t0; T["copy"] += t2 - t1; T["sync"] += t3 - t2; T["npcopy"] += t4 - t3; T["parse"] += t5 - t4
    n += 1
t0 = pc()
for a in acts:
    env.step(int(a))
T["whole"] = pc() - t0
print({k: round(1e6 * v / n, 2) for k, v in T.items()}, "us per step; obs", o.shape, o.dtype)
