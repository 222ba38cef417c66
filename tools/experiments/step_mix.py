#!/usr/bin/env python3
"""One step per launch, state only, 65 536 environments: which part of the C4 mix costs what.
Pools: Level 0 only (all families), Levels 1-4 only, and the C4 shard; N_pad as the pool needs it and forced to 32."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pushworld_amd import _capi  # noqa: E402
from pushworld_amd import benchmark_data as bd  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402
from tools.experiments.step_ab import timed  # noqa: E402

B = 65536
l0 = list(bd.level0_texts().values())
hi = []
for lv in (1, 2, 3, 4):
    for p in bd.level_paths(lv):
        with open(p) as f:
            hi.append(f.read())
print("library:", os.path.basename(_capi.LIB_PATH))
for name, texts in (("Level 0 only (%d puzzles)" % len(l0), l0), ("Levels 1-4 only (%d puzzles)" % len(hi), hi),
                    ("Level 1-4 without objects > 8x8", None), ("both", l0 + hi)):
    if texts is None:
        texts = []
        for t in hi:
            pp = _capi.ParsedPuzzle(t)
            ok = True
            for cells in pp.object_cells:
                c = np.array(cells)
                ok = ok and (c[:, 0].max() - c[:, 0].min() < 8) and (c[:, 1].max() - c[:, 1].min() < 8)
            if ok:
                texts.append(t)
        name += " (%d puzzles)" % len(texts)
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    ids = np.sort(np.random.default_rng(0).integers(0, len(texts), B))
    vec = VecPushWorld(pset, B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True)
    vec.engine.set_option("step_lds_tables", 2)
    vec.reset()
    g = torch.Generator(device=vec.device).manual_seed(1)
    acts = torch.randint(0, 4, (64, B), generator=g, device=vec.device, dtype=torch.uint8)
    it = [0]

    def one():
        vec.step(acts[it[0] % 64])
        it[0] += 1

    for _ in range(100):
        one()
    med, mn = timed(one, 100)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300):
        one()
    b.record()
    torch.cuda.synchronize()
    print("%-44s N_pad %2d  step us med %7.2f  min %7.2f   300 back to back: %7.2f us each" %
          (name, vec.engine.np, med, mn, a.elapsed_time(b) / 300 * 1e3), flush=True)
    del vec
