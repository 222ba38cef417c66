#!/usr/bin/env python3
"""Can a process that drew only slow-class observation buffers recover?  Rounds of 10 candidate allocations (all
alive within a round), between rounds everything is freed and a spacer of growing size is left allocated, so the
next round's candidates land elsewhere.  Prints the tuner's best launch time per candidate (ms)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402

B = 65536
paths = bench.level1_paths()
ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                   border_width=1, observation="uint8", autoreset=True, tune=False)
vec.reset()
eng = vec.engine
spacers = []
for rnd, spacer_gb in enumerate((0, 0, 7, 33, 1.3)):
    if spacer_gb:
        spacers.append(torch.empty(int(spacer_gb * (1 << 30)), dtype=torch.uint8, device=vec.device))
    cands, ms = [], []
    for k in range(10):
        storage, view = eng.alloc_obs(B)
        cands.append(storage)
        eng.tune_render(vec.puzzle_id, vec.pos, storage)
        ms.append(round(eng.get_option("tuned_ns") * 1e-6, 4))
    print("round %d (spacers held: %s GB): %s" % (rnd, [round(s.numel() / (1 << 30), 1) for s in spacers], ms), flush=True)
    del cands, storage, view
    torch.cuda.empty_cache()
