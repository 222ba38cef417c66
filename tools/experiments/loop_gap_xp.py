#!/usr/bin/env python3
"""Why is the render ~2 % slower inside the step / render loop than in the tuner?  Same buffer, same launch configuration:
(a) renders back to back on the reset states (what the tuner measures), (b) renders back to back on the states after 300
random steps (objects spread over the boards), (c) step + render alternating (the bench loop), (d) as (c) with the step
kernel replaced by an idle gap of the same length -- kernel durations from the library's own events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402


def main():
    B = 65536
    paths = bench.level1_paths()
    pool = [PushWorldPuzzle(p) for p in paths]
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, observation="uint8", pixels_per_cell=3, border_width=1, device=0,
                       autoreset=True)
    eng = vec.engine
    print("candidates", [round(c, 4) for c in vec.tuned_candidates_ms], "tuned", vec.tuned_ms, "config", vec.tuned_config)
    g = torch.Generator(device=vec.device).manual_seed(1)
    acts = torch.randint(0, 4, (512, B), generator=g, device=vec.device, dtype=torch.uint8)

    def renders(n, label):
        for _ in range(5):
            vec.render()
        eng.profile_render(n)
        for _ in range(n):
            vec.render()
        ms = np.array(eng.profile_read())
        print(f"{label:60s} mean {ms.mean():.4f}  median {np.median(ms):.4f}  min {ms.min():.4f} ms", flush=True)

    vec.reset()
    renders(200, "(a) back to back, reset states")
    for t in range(300):
        eng.step(vec.puzzle_id, acts[t], vec.pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated, vec.flags)
    renders(200, "(b) back to back, states after 300 random steps")
    eng.profile_render(400)
    for t in range(400):
        eng.step_render(vec.puzzle_id, acts[t % 512], vec.pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated,
                        vec._obs_storage, vec.flags)
    ms = np.array(eng.profile_read())
    print(f"{'(c) step + render alternating (pw_step_render)':60s} mean {ms.mean():.4f}  median {np.median(ms):.4f}  min {ms.min():.4f} ms")
    # (d) a small unrelated kernel between the renders instead of the step
    filler = torch.zeros(1 << 16, device=vec.device)
    eng.profile_render(400)
    for t in range(400):
        filler.add_(1.0)
        vec.render()
    ms = np.array(eng.profile_read())
    print(f"{'(d) tiny torch kernel + render alternating':60s} mean {ms.mean():.4f}  median {np.median(ms):.4f}  min {ms.min():.4f} ms")
    renders(2000, "(e) 2000 renders back to back (sustained, ~1.1 s)")
    renders(200, "(f) 200 more right after")


if __name__ == "__main__":
    main()
