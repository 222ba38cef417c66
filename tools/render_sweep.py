#!/usr/bin/env python3
"""Render-kernel bandwidth for several (ppc, bw, dtype) settings on the Level-1 mix."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402


def main():
    # --torch: the observation buffer from torch's allocator, tuned in place (tune_allocations = 0) instead of library-owned
    # candidates; --chunk MB: physical chunk size of the library-owned buffers
    owned = "--torch" not in sys.argv
    opts = {"obs_chunk_mb": int(sys.argv[sys.argv.index("--chunk") + 1])} if "--chunk" in sys.argv else None
    print("observation buffers:", "library-owned (pw_obs_alloc_tuned)" if owned else "torch allocator, tuned in place", opts or "")
    paths = bench.level1_paths()
    pool = [PushWorldPuzzle(p) for p in paths]
    for ppc, bw, obs, B in [(3, 1, "uint8", 65536), (3, 1, "float32", 32768), (8, 2, "uint8", 16384),
                            (20, 2, "uint8", 4096), (20, 2, "float32", 2048)]:
        ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
        vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, border_width=bw, pixels_per_cell=ppc,
                           observation=obs, device=0, autoreset=True, tune_allocations=None if owned else 0,
                           engine_options=opts)
        vec.reset()
        torch.cuda.synchronize()
        if "--settle" in sys.argv:  # memory the allocator just gave back may still be being cleared by the driver
            import time
            time.sleep(1.0)
            for _ in range(20):
                vec.render()
            torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(24 if "--settle" in sys.argv else 8)]
        for a, b in evs:
            a.record()
            vec.render()
            b.record()
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in evs])
        nbytes = B * vec.engine.obs_bytes
        print(f"ppc={ppc:2d} bw={bw} {obs:8s} B={B:6d} obs={vec.engine.obs_bytes / 1e6:8.3f} MB  {ms:8.3f} ms  "
              f"{nbytes / ms / 1e6:8.1f} GB/s  {B / ms * 1e3:12.0f} renders/s   tuned {vec.tuned_ms:.4f} ms, candidates "
              f"{[round(c, 4) for c in vec.tuned_candidates_ms]}, config {vec.tuned_config}", flush=True)
        del vec
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
