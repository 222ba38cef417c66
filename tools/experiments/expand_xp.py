#!/usr/bin/env python3
"""Round-4 experiment: launch variants of pw_expand4_v2_kernel (tile order, prefetch, workgroups per CU) against
pw_expand4_lane_kernel on cache-busting frontiers of the three BASELINE puzzles.  Prints one line per variant."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import config_suite as cs  # noqa: E402


GROUPS = tuple(int(v) for v in next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--gp=")), "2,3,4,8").split(","))


def main():
    puzzles = [a for a in sys.argv[1:] if not a.startswith("--")] or ["level1/2 Obstacle.pwp", "level2/Pull Dont Push.pwp", "level4/Four Pistons.pwp", "level3/Armor.pwp",
                               "level1/Pull Up.pwp"]
    for rel in puzzles:
        pz, st_host, exhausted, distinct = cs.c5_frontier(rel, 4_000_000)
        F, N = st_host.shape
        eng = pz._engine()
        dev = eng.device
        per = 20 * N + 20
        nbuf = max(1, -(-cs.CACHE_BUST_BYTES // (F * per)))
        sets = [(torch.as_tensor(st_host).to(dev), torch.empty((F, 4, N), dtype=torch.int32, device=dev),
                 torch.empty((F, 4), dtype=torch.int32, device=dev), torch.empty((F, 4), dtype=torch.uint8, device=dev)) for _ in range(nbuf)]
        ref = None
        quick = "--quick" in sys.argv  # only the non-temporal-store kernels, no v1
        variants = [] if quick else [("v1 (tables through L1)", {"expand_lds_tables": 2})]
        if "--v1runs" in sys.argv:
            variants = [("v1 (tables through L1)", {"expand_lds_tables": 2}), ("v1, all four actions per pass, non-temporal", {"expand_lds_tables": 3})]
        orders = (0,) if quick else ((0, 1, 2) if "--orders" in sys.argv else (0, 2))  # (2: plain stores instead of non-temporal ones)
        if "--wgwaves" in sys.argv:  # lone workgroups per CU (tables beyond ~70 KB): wavefronts per workgroup
            variants = [(f"v2 automatic, at most {w or 8} wavefronts per workgroup", {"expand_lds_tables": 0, "expand_tile_order": 0, "expand_prefetch": -1,
                                                                                 "expand_groups_per_cu": 0, "expand_wg_waves": w}) for w in (4, 0)]
        for order in (() if ("--v1runs" in sys.argv or "--wgwaves" in sys.argv) else orders):
            for pre in ((0, 2) if quick else (-1, 0, 2)):  # 0: stores at the end of a tile (kPipe 0), 2: one tile late (kPipe 1)
                for gp in ((0,) if pre < 0 else GROUPS):
                    variants.append((f"v2 order {order} prefetch {'auto' if pre < 0 else pre} groups/CU {gp or 'auto'}",
                                     {"expand_lds_tables": 0, "expand_tile_order": order, "expand_prefetch": pre, "expand_groups_per_cu": gp}))
        for name, opts in variants:
            for k, v in opts.items():
                eng.set_option(k, v)
            it = [0]

            def one():
                s = sets[it[0] % nbuf]
                it[0] += 1
                eng.expand4(0, s[0], s[1], s[2], s[3])

            reps = max(8, 2 * nbuf)
            dt = cs.wall(one, reps, nbuf)
            ms = cs.launch_ms(eng, one, reps)
            got = [t.clone() for t in sets[0][1:]]
            if ref is None:
                ref = got
            same = all(torch.equal(a, b) for a, b in zip(got, ref))
            gbs = F * per / (ms.mean() * 1e-3) / 1e9
            print(f"{rel:28s} N={N:2d} F={F} {name:42s} {ms.mean():7.4f} ms  {F / ms.mean() * 1e3:10.3e} parents/s  {gbs:7.1f} GB/s = {gbs / 8000:.3f} of peak"
                  f"  wall {dt * 1e3:7.4f} ms  {'same' if same else 'DIFFERENT RESULTS'}", flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
