#!/usr/bin/env python3
"""Render kernel with EVERY page loading its static chunks (PW_PAGE_XP=1 build via PUSHWORLD_AMD_LIB) against the
production build: does equalising the pages' latencies smooth the write fronts on slow-class buffers?
10 candidate buffers; per buffer the tuner's best time and the times of a few fixed launch configurations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pushworld_amd import _capi  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402

B = 65536
paths = bench.level1_paths()
ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                   border_width=1, observation="uint8", autoreset=True, tune=False)
vec.reset()
eng = vec.engine
CFGS = [("runs64+7K", (2, 6, 7)), ("eighths+0", (1, 0, 0)), ("eighths+4K", (1, 0, 4)), ("eighths+7K", (1, 0, 7)), ("quarters+5K", (1, 1, 5)),
        ("quarters+7K", (1, 1, 7))]


def timed(storage, cfg, reps=8):
    for k, v in zip(("page_order", "page_run_log2", "page_lds_pad_kb"), cfg):
        eng.set_option(k, v)
    for _ in range(2):
        eng.render(vec.puzzle_id, vec.pos, storage)
    eng.profile_render(reps)
    for _ in range(reps):
        eng.render(vec.puzzle_id, vec.pos, storage)
    ms = np.array(eng.profile_read())
    eng.profile_render(0)
    return float(np.median(ms))


print("library:", os.path.basename(_capi.LIB_PATH))
print("%-10s %8s  " % ("buffer", "tuned") + " ".join("%12s" % n for n, _ in CFGS))
keep = []
for k in range(8):
    storage, view = eng.alloc_obs(B)
    keep.append(storage)
    eng.tune_render(vec.puzzle_id, vec.pos, storage)
    tuned = eng.get_option("tuned_ns") * 1e-6
    print("%-10s %8.4f  " % ("#%d" % k, tuned) + " ".join("%12.4f" % timed(storage, c) for _, c in CFGS), flush=True)
