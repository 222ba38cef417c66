"""Render bandwidth at the batch sizes RL code actually uses, reference-default observation (ppc 20, float32)
and ppc 3: is one workgroup per environment enough parallelism?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pushworld_amd.puzzle import PushWorldPuzzle
from pushworld_amd.vec_env import VecPushWorld
pool = [PushWorldPuzzle(p) for p in bench.level1_paths()]
for ppc, bw, obs in [(20, 2, "float32"), (20, 2, "uint8"), (8, 2, "uint8"), (3, 1, "float32"), (3, 1, "uint8")]:
    for B in (64, 256, 1024, 4096):
        ids = (np.arange(B, dtype=np.int64) * len(pool)) // B
        vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, border_width=bw, pixels_per_cell=ppc, observation=obs, autoreset=True)
        vec.reset()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in evs:
            a.record(); vec.render(); b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
        print(f"ppc={ppc:2d} {obs:8s} B={B:5d} {vec.engine.render_kernel:26s} {ms:8.3f} ms {B * vec.engine.obs_bytes / ms / 1e6:8.1f} GB/s", flush=True)
        del vec
