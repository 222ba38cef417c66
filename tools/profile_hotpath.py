#!/usr/bin/env python3
"""Small driver for rocprofv3: N step+render passes of the C3 workload, nothing else timed.

    rocprofv3 --kernel-trace --stats -d out -o name -- python tools/profile_hotpath.py --steps 20
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o name -- python tools/profile_hotpath.py
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--ppc", type=int, default=3)
    ap.add_argument("--bw", type=int, default=1)
    ap.add_argument("--obs", default="uint8")
    ap.add_argument("--config", default="c3", choices=("c3", "c4"))
    args = ap.parse_args()
    B = args.envs
    if args.config == "c4":  # rank 0's shard of the 8-rank assignment, exactly as bench.py --config c4 builds it
        wl = argparse.Namespace(envs_per_gpu=B, obs=args.obs, config="c4", max_steps=200, bw=args.bw, ppc=args.ppc,
                                tune_allocations=None)
        vec = bench.build_workload(wl, 0, 8, 0)["vec"]
    else:
        paths = bench.level1_paths()
        ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
        # fused=True: the bench's path (pw_step_render: the step kernel leaves the page records, no pre-pass); the
        # tuner's trial launches at reset() run under their own kernel symbol (pw_render_page_kernel<T, 1>)
        vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, border_width=args.bw,
                           pixels_per_cell=args.ppc, observation=args.obs, device=0, autoreset=True, fused=True)
    vec.reset()
    gen = torch.Generator(device=vec.device)
    gen.manual_seed(1)
    acts = torch.randint(0, 4, (args.steps, B), generator=gen, device=vec.device, dtype=torch.uint8)
    torch.cuda.synchronize()
    for t in range(args.steps):
        vec.step(acts[t])
    torch.cuda.synchronize()
    eng = vec.engine
    print("done", args.steps, "steps of", B, "envs; obs bytes/env", eng.obs_bytes, "render launch config",
          [eng.get_option(k) for k in ("page_order", "page_run_log2", "page_lds_pad_kb")], "tuned", vec.tuned_config)


if __name__ == "__main__":
    main()
