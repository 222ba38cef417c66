#!/usr/bin/env python3
"""PW_OPT_STEP_TABLES A/B (state only, lane-group kernel): overlap tables for no puzzle / for the puzzles with movables
beyond 8 x 8 only (kernel with both paths) / automatic = the default (every puzzle of a set that has such a movable:
the table-only kernel; none for sets without).  One step per launch (median / min microseconds over --reps launches, HIP
events) and 64-step rollouts.  Workloads: C2, C3, the C4 shard, Levels 1-4 only, Levels 1-4 without the big-object puzzles,
65 536 copies of Mind The Gap (the slowest puzzle of profiles/r02_step_xp.txt section 7)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pushworld_amd import _capi  # noqa: E402
from pushworld_amd import benchmark_data as bd  # noqa: E402
from pushworld_amd.sharding import c4_global_puzzle_ids, shard_puzzle_ids  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402


def timed(fn, n):
    for _ in range(5):
        fn()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    return float(np.median(t)) * 1e3, float(t.min()) * 1e3


def level_texts(levels):
    out = []
    for lv in levels:
        for p in bd.level_paths(lv):
            with open(p) as f:
                out.append((p, f.read()))
    return out


def is_big(parsed):
    return any(max(c[0] for c in cells) >= 8 or max(c[1] for c in cells) >= 8 for cells in parsed.object_cells)


def workloads(which, B=65536):
    if "c2" in which:
        l0 = bd.load_level0(("base",), "train", 1)
        yield "C2 4096 x one L0 puzzle", l0, 4096, np.zeros(4096, np.int64)
    if "c3" in which:
        from pushworld_amd.puzzle import PushWorldPuzzle
        paths = bench.level1_paths()
        yield "C3 65536 Level-1", [PushWorldPuzzle(p) for p in paths], B, (np.arange(B, dtype=np.int64) * len(paths)) // B
    if "l0" in which:
        l0 = [_capi.ParsedPuzzle(t) for t in bd.level0_texts(("all",), "train", 2000).values()]
        yield "65536 over 2000 L0 'all'", _capi.PuzzleSet(l0, 0), B, (np.arange(B, dtype=np.int64) * len(l0)) // B
    if "l0tiny" in which:  # the 5 x 5 families: every puzzle fits 8 x 8 with its border (pw_step_board_kernel)
        l0 = [_capi.ParsedPuzzle(t) for t in bd.level0_texts(("base", "walls", "goals", "obstacles", "shapes"), "train", 400).values()]
        yield "%d 5x5 Level-0 puzzles" % len(l0), _capi.PuzzleSet(l0, 0), B, (np.arange(B, dtype=np.int64) * len(l0)) // B
    hi = level_texts((1, 2, 3, 4))
    if "c4" in which:
        texts = list(bd.level0_texts().values())
        n_l0 = len(texts)
        texts += [t for _, t in hi]
        ids = np.sort(shard_puzzle_ids(c4_global_puzzle_ids(8 * B, n_l0, len(hi), 100), 0, 8))
        yield "C4 shard 14223 puzzles", _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0), B, ids
    parsed = [_capi.ParsedPuzzle(t) for _, t in hi]
    if "l14" in which:
        yield "Levels 1-4 only (223)", _capi.PuzzleSet(parsed, 0), B, (np.arange(B, dtype=np.int64) * len(parsed)) // B
    if "nobig" in which:
        small = [p for p in parsed if not is_big(p)]
        pad = [p for p in parsed if p.num_movables > 16][:1]  # keeps N_pad 32 like the full pool
        pool = small + [p for p in pad if p not in small]
        n = len(small)
        yield "Levels 1-4 without big (%d)" % n, _capi.PuzzleSet(pool, 0), B, (np.arange(B, dtype=np.int64) * n) // B
    if "gap" in which:
        k = [i for i, (p, _) in enumerate(hi) if p.endswith("Mind The Gap.pwp")][0]
        pad = [p for p in parsed if p.num_movables > 16][:1]
        yield "Mind The Gap x 65536", _capi.PuzzleSet([parsed[k]] + pad, 0), B, np.zeros(B, np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--which", default="c2,c3,c4,l14,nobig,gap")
    ap.add_argument("--modes", default="none+fwd,auto+fwd,auto,all+fwd,all,all+narrow")
    ap.add_argument("--batch", type=int, default=65536, help="environments of every workload but C2")
    args = ap.parse_args()
    print("%-30s %-12s %9s %9s %13s %15s %10s" % ("workload", "tables", "step med", "step min", "rollout64 us", "rollout steps/s", "table KB"))
    for name, pool, B, ids in workloads(args.which.split(","), args.batch):
        for mode in args.modes.split(","):
            opts = {"step_tables": mode.split("+")[0]}
            if "+16lanes" in mode:  # N_pad 32: 16-lane groups for every environment
                opts["step_narrow_groups"] = 2
            elif "+narrow" in mode:
                opts["step_narrow_groups"] = 1
            if "+fwd" in mode:   # (VecPushWorld picks the reverse order by itself when the expensive puzzles come last)
                opts["step_block_order"] = "forward"
            if "+lane" in mode and "+nolane" not in mode:  # one lane per environment (table-driven when every puzzle has tables)
                opts["step_kernel"] = "lane"
            if "+noboards" in mode:  # sets of 8 x 8 puzzles: the lane groups instead of the whole-grid boards in registers
                opts["step_boards"] = "never"
            if "+nolane" in mode:  # lane groups whatever the batch size
                opts["step_lane_batch"] = "never"
            if "+rev" in mode:
                opts["step_block_order"] = "reverse"
            vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True, engine_options=opts)
            vec.reset()
            g = torch.Generator(device=vec.device).manual_seed(1)
            acts = torch.randint(0, 4, (64, B), generator=g, device=vec.device, dtype=torch.uint8)
            it = [0]

            def one():
                vec.step(acts[it[0] % 64])
                it[0] += 1

            med, mn = timed(one, args.reps)
            rmed, _ = timed(lambda: vec.rollout(acts), max(10, args.reps // 5))
            print("%-30s %-12s %9.2f %9.2f %13.1f %15.3e %10d%s" % (name, mode, med, mn, rmed, 64 * B / (rmed * 1e-6),
                                                                   vec.engine.get_option("step_table_bytes") >> 10,
                                                                   "  reverse" if vec.engine.get_option("step_block_order") else ""), flush=True)
            del vec


if __name__ == "__main__":
    main()
