"""State-only stepping of a C4 shard / the C3 set, bound (pw_batch_bind) against unbound: one step per launch and 64-step
launches, HIP events around runs of launches on torch's stream (the engine launches there).

    python tools/bench_bind.py [--config c4|c3] [--batch 65536] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build(config, B, bind, opts):
    from pushworld_amd import _capi
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.sharding import c4_global_puzzle_ids, shard_puzzle_ids
    from pushworld_amd.vec_env import VecPushWorld

    if config == "c2":  # 4 096 (or --batch) copies of one Level-0 puzzle (BASELINE config 2)
        texts = [next(iter(bd.level0_texts(("base",), "train", 1).values()))]
        ids = np.zeros(B, dtype=np.int64)
    elif config == "c3":
        texts = [open(p).read() for p in bd.level_paths(1)]
        ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
    else:
        texts = list(bd.level0_texts().values())
        n_l0 = len(texts)
        for lv in (1, 2, 3, 4):
            for p in bd.level_paths(lv):
                texts.append(open(p).read())
        ids = np.sort(shard_puzzle_ids(c4_global_puzzle_ids(8 * B, n_l0, len(texts) - n_l0, 100), 0, 8))
        if config == "c4hi":
            ids = ids[ids >= n_l0]
            ids = np.sort(np.concatenate([ids, ids])[:B])
        elif config == "c4lo":
            ids = ids[ids < n_l0]
            ids = np.sort(np.concatenate([ids, ids])[:B])
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    vec = VecPushWorld(pset, B, puzzle_ids=ids, max_steps=100, observation=None, device=0, autoreset=True, bind=bind, engine_options=opts)
    vec.reset()
    return vec


def timed(fn, n, reps=5):
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(best)), float(min(best))


def per_puzzle(args):
    from pushworld_amd import _capi
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.vec_env import VecPushWorld

    names, texts = [], []
    for lv in (1, 2, 3, 4):
        for p in bd.level_paths(lv):
            names.append(f"level{lv}/" + os.path.splitext(os.path.basename(p))[0])
            texts.append(open(p).read())
    B = args.per_puzzle
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    vec = VecPushWorld(pset, B, puzzle_ids=np.zeros(B, np.int64), max_steps=100, observation=None, device=0, autoreset=True, bind=True,
                       engine_options={"bind_spread": args.spread} if args.spread else {})
    hdr = np.frombuffer(pset.headers(), np.uint8).reshape(-1, _capi.PUZZLE_HEADER_BYTES)
    acts = torch.as_tensor(np.random.default_rng(0).integers(0, 4, size=(64, B), dtype=np.uint8)).cuda()
    rows = []
    for i, name in enumerate(names):
        vec.set_puzzle_ids(np.full(B, i, np.int64))
        vec.reset()
        vec.rollout(acts)
        med, _ = timed(lambda: vec.rollout(acts), 3, reps=3)
        rows.append((med / 64, name, int(hdr[i, _capi.PUZZLE_HEADER_N_OFFSET]), vec.bound_info["bound_envs"]))
    rows.sort(reverse=True)
    us = np.array([r[0] for r in rows])
    print(f"{B} environments of one puzzle, 64-step launches, us per step: median {np.median(us):.2f} p90 {np.percentile(us, 90):.2f} max {us.max():.2f}")
    for r in rows[:args.rows]:
        print(f"  {r[0]:7.2f} us/step  N={r[2]:2d} bound={r[3]}  {r[1]}")
    if args.json:
        with open(args.json, "w") as f:
            json.dump([{"us_per_step": round(r[0], 3), "puzzle": r[1], "movables": r[2], "bound_envs": r[3]} for r in rows], f, indent=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--json", default=None)
    ap.add_argument("--variants", default="unbound,bound,bound-split")
    ap.add_argument("--per-puzzle", type=int, default=0, help="environments of ONE Level 1-4 puzzle at a time (bound): microseconds per step of 64-step launches")
    ap.add_argument("--rows", type=int, default=25, help="--per-puzzle: lines printed (slowest first)")
    ap.add_argument("--spread", type=int, default=0, help="PW_OPT_BIND_SPREAD of the per-puzzle runs")
    args = ap.parse_args()
    if args.per_puzzle:
        return per_puzzle(args)
    B = args.batch
    rng = np.random.default_rng(0)
    acts1 = torch.as_tensor(rng.integers(0, 4, size=(256, B), dtype=np.uint8)).cuda()
    acts64 = acts1[:64].contiguous()
    out = {"config": args.config, "batch": B, "device": torch.cuda.get_device_name(0)}
    for name in args.variants.split(","):
        bind = name.startswith("bound")
        opts = {"bind_fused": 2} if name == "bound-split" else {}
        if "noboards" in name:  # sets of 8 x 8 puzzles: not the whole-grid board kernel
            opts["step_boards"] = "never"
        if name.endswith("-noquad"):
            opts["step_quad16"] = "never"
        if "kb16" in name:
            opts["bind_max_kb"] = 16
        for k in (1, 2, 3):
            if f"lanes{k}" in name:
                opts["bind_lanes"] = k
        for k in (1, 2, 3, 4, 5):
            if f"spread{k}" in name:
                opts["bind_spread"] = k
            if f"list{k}" in name:  # the listed environments only (pw_step_mseg_kernel)
                opts["bind_spread"] = 16 * k
        vec = build(args.config, B, bind, opts)
        for k in range(64):
            vec.step(acts1[k])
        step_med, step_min = timed(lambda: [vec.step(acts1[k]) for k in range(128)], 1, reps=7)
        step_med, step_min = step_med / 128, step_min / 128
        vec.rollout(acts64)
        roll_med, roll_min = timed(lambda: vec.rollout(acts64), 10, reps=5)
        out[name] = {"step_us": round(step_med, 3), "step_us_min": round(step_min, 3), "rollout64_us": round(roll_med, 2),
                     "rollout64_us_min": round(roll_min, 2), "rollout_env_steps_per_s": round(B * 64 / (roll_med * 1e-6), -6),
                     "bound_info": vec.bound_info}
        print(name, out[name], flush=True)
        del vec
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
