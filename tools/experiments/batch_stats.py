#!/usr/bin/env python3
"""Round-4 probe: state counts of pw_search_batch by verdict on generated Level-0 puzzles, and its rate at several caps."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pushworld_amd import _capi, generate
from pushworld_amd.search import search_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4000
pset, grids, dims = generate.generate_level0_set(n, device=0, random_seed=21)
eng = _capi.Engine(pset, None, 3, 1, _capi.OBS_U8)
if "--groups" in sys.argv:  # A/B of PW_OPT_SEARCH_BATCH_GROUPS_PER_CU
    for cap in (1 << 14, 1 << 16, 1 << 18):
        for g in (1, 2, 3, 4, 6, 8):
            eng.set_option("search_batch_groups_per_cu", g)
            search_batch(eng, None, max_states=cap)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                v, pl, ns = search_batch(eng, None, max_states=cap)
                best = min(best, time.perf_counter() - t0)
            print(f"cap {cap:8d} groups/CU {g}: {n / best:9.0f} puzzles/s ({best * 1e3:7.2f} ms)  {ns.sum() / best:10.3e} states/s", flush=True)
    eng.set_option("search_batch_groups_per_cu", 0)
    sys.exit(0)
for cap in (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 21):
    search_batch(eng, None, max_states=cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v, pl, ns = search_batch(eng, None, max_states=cap)
    dt = time.perf_counter() - t0
    print(f"cap {cap:8d}: {n / dt:9.0f} puzzles/s ({dt * 1e3:7.2f} ms)  verdicts {np.bincount(v, minlength=4).tolist()}  states sum {int(ns.sum()):11d}  "
          f"{ns.sum() / dt:10.3e} states/s", flush=True)
    if cap == 1 << 21:
        for name, code in (("solved", 1), ("unsolvable", 0), ("unknown", 2)):
            x = np.sort(ns[v == code])
            if len(x):
                q = [int(x[int(p * (len(x) - 1))]) for p in (0.1, 0.5, 0.9, 0.99, 1.0)]
                print(f"   {name:10s} n={len(x):5d} states p10/p50/p90/p99/max {q}  plan_len median {int(np.median(pl[v == code])) if code == 1 else -1}")
