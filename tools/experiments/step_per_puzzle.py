#!/usr/bin/env python3
"""One step per launch, 65 536 copies of ONE puzzle, for every Level 1-4 puzzle: which puzzles are slow?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pushworld_amd import _capi  # noqa: E402
from pushworld_amd import benchmark_data as bd  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402

B = 65536
rows = []
paths = [p for lv in (1, 2, 3, 4) for p in bd.level_paths(lv)]
texts = [open(p).read() for p in paths]
pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 4, (64, B), generator=g, device="cuda", dtype=torch.uint8)
for i, p in enumerate(paths):
    pp = _capi.ParsedPuzzle(texts[i])
    dims = []
    for cells in pp.object_cells:
        c = np.array(cells)
        dims.append((int(c[:, 0].max() - c[:, 0].min() + 1), int(c[:, 1].max() - c[:, 1].min() + 1)))
    vec = VecPushWorld(pset, B, puzzle_ids=np.full(B, i), max_steps=200, observation=None, autoreset=True)
    vec.engine.set_option("step_lds_tables", 2)
    vec.reset()
    for k in range(64):
        vec.step(acts[k])
    best = 1e9
    for rep in range(4):  # best of 4 x 128 steps (allocator / clock hiccups are ~100 ms)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(128):
            vec.step(acts[k % 64])
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 128 * 1e3)
    rows.append((best, os.path.relpath(p, os.path.dirname(os.path.dirname(p))), len(dims),
                 max(d[0] for d in dims), max(d[1] for d in dims), dims[0]))
    del vec
rows.sort(reverse=True)
print("us/step  puzzle  N  max w  max h  agent (w, h)    [N_pad 32 for all: one set]")
for r in rows[:40]:
    print("%7.2f  %-40s N=%2d  w<=%2d h<=%2d  agent %s" % r)
print("...")
for r in rows[-5:]:
    print("%7.2f  %-40s N=%2d  w<=%2d h<=%2d  agent %s" % r)
print("mean %.2f us" % np.mean([r[0] for r in rows]))
