#!/usr/bin/env python3
"""HBM write/copy bandwidth probes next to the render kernel (run on the GPU box).
Prints one line per probe: name, ms, GB/s."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in evs])
    return float(np.median(ms)), float(ms.min())


def main():
    paths = bench.level1_paths()
    B = 65536
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, border_width=1,
                       pixels_per_cell=3, observation="uint8", device=0, autoreset=True)
    vec.reset()
    st = vec._obs_storage
    nbytes = st.numel()
    print("buffer bytes", nbytes)
    for name, fn, traffic in [
        ("fill_0", lambda: st.fill_(0), nbytes),
        ("fill_90", lambda: st.fill_(90), nbytes),
        ("copy_half", lambda: st[: B // 2].copy_(st[B // 2:]), nbytes),
        ("render", lambda: vec.engine.render(vec.puzzle_id, vec.pos, st), B * vec.engine.obs_bytes),
    ]:
        med, mn = timeit(fn)
        print(f"{name:10s} median {med:.4f} ms  min {mn:.4f} ms  {traffic / med / 1e6:.1f} GB/s (median)")
    x = torch.empty((nbytes // 4,), dtype=torch.float32, device=st.device)
    y = torch.empty_like(x)
    for name, fn, traffic in [
        ("f32_fill", lambda: x.fill_(1.5), nbytes),
        ("f32_copy", lambda: y.copy_(x), 2 * nbytes),
        ("f32_add", lambda: torch.add(x, 1.0, out=y), 2 * nbytes),
    ]:
        med, mn = timeit(fn)
        print(f"{name:10s} median {med:.4f} ms  min {mn:.4f} ms  {traffic / med / 1e6:.1f} GB/s (median)")


if __name__ == "__main__":
    main()
