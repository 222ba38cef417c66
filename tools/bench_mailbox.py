#!/usr/bin/env python3
"""C2 (4 096 copies of one Level-0 puzzle, state only) one step at a time: what a step costs the HOST that needs its verdicts
before it chooses the next actions -- pw_step launches against the resident kernel (pw_mailbox_*).

    python tools/bench_mailbox.py [--envs 4096] [--steps 20000] > gpurun_out/mailbox.json

Lines of the record (us per step; env-steps/s = envs / that):
  launch_async           pw_step launches queued back to back, one synchronisation at the end (no verdict reaches the host)
  launch_sync            pw_step + stream synchronisation per step (verdicts still on the device)
  launch_sync_verdicts   ... + the three verdict arrays copied to the host (what gym-style host loops do)
  mailbox_sync_dev/_host pw_mailbox_run ahead 1: every step waits for the one before; verdicts in pinned host memory; actions in
                         device memory / in host memory (copied to a pinned slot, read by the kernel across the link)
  mailbox_ahead4/8_dev   ... up to 4 / 8 steps in flight
  mailbox_python_step    Mailbox.step() from Python (ctypes + numpy views per call)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--pool", default="c2", help="c2 (one 5 x 5 Level-0 puzzle) or level1")
    ap.add_argument("--modes", default="3", help="PW_OPT_MAILBOX_MODE values to run, e.g. 0,1,2,4,6")
    args = ap.parse_args()
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    B, T = args.envs, args.steps
    if args.pool == "level1":  # the 68 Level-1 puzzles (N_pad 16: lane_step inside the resident kernel, lane groups for pw_step)
        import bench
        member = "level1 (68 puzzles, environments grouped by puzzle)"
        puzzles = [PushWorldPuzzle(p) for p in bench.level1_paths()]
        ids = (np.arange(B, dtype=np.int64) * len(puzzles)) // B
        vec = VecPushWorld(puzzles, B, puzzle_ids=ids, max_steps=100, observation=None, device=0, autoreset=True)
    else:
        member = "level0/base/train/level_0_base_train_0.pwp"
        text = bd.level0_texts()[member] if member in bd.level0_texts() else next(iter(bd.level0_texts().values()))
        vec = VecPushWorld([PushWorldPuzzle(text=text)], B, max_steps=100, observation=None, device=0, autoreset=True)
    vec.reset()
    acts = np.random.default_rng(0).integers(0, 4, size=(T, B), dtype=np.uint8)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    torch.cuda.synchronize()
    out = {"envs": B, "steps": T, "puzzle": member, "unit": "us per step"}

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    def launch_async(n):
        for t in range(n):
            vec.step(acts_dev[t])

    def launch_sync(n):
        for t in range(n):
            vec.step(acts_dev[t])
            torch.cuda.current_stream().synchronize()

    def launch_sync_verdicts(n):
        for t in range(n):
            _, r, te, tr = vec.step(acts_dev[t])
            r.cpu(), te.cpu(), tr.cpu()

    n_launch = min(T, 5000)
    launch_async(200)
    out["launch_async"] = timed(launch_async, n_launch)
    out["launch_sync"] = timed(launch_sync, n_launch)
    out["launch_sync_verdicts"] = timed(launch_sync_verdicts, min(T, 2000))

    modes = [int(v) for v in args.modes.split(",")]
    for mode in modes:
        vec.engine.set_option("mailbox_mode", mode)
        tag = "" if len(modes) == 1 else f"mode{mode}_"
        mb = vec.mailbox(ring=8)
        mb.run(acts_dev[:500], 1)
        for name, arr, ahead in (("mailbox_sync_dev", acts_dev, 1), ("mailbox_sync_host", acts, 1), ("mailbox_ahead4_dev", acts_dev, 4),
                                 ("mailbox_ahead8_dev", acts_dev, 8), ("mailbox_ahead8_host", acts, 8)):
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                mb.run(arr, ahead)
                dt = (time.perf_counter() - t0) / T * 1e6
                best = dt if best is None else min(best, dt)
            out[tag + name] = best
        n_py = min(T, 5000)
        t0 = time.perf_counter()
        for t in range(n_py):
            mb.step(acts[t])
        out[tag + "mailbox_python_step"] = (time.perf_counter() - t0) / n_py * 1e6
        mb.close()
        # the per-phase clock of wavefront 0 (mode bit 4: it slows that wavefront, hence every step -- its own runs)
        vec.engine.set_option("mailbox_mode", mode | 16)
        # ... the synchronous cadence (the profile of a host that waits for every step)
        mb = vec.mailbox(ring=8)
        mb.run(acts_dev, 1)
        out[tag + "wavefront0_profile_sync_dev"] = mb.close(profile=True)
        # ... and eight steps in flight alone
        mb = vec.mailbox(ring=8)
        mb.run(acts_dev, 8)
        out[tag + "wavefront0_profile_ahead8_dev"] = mb.close(profile=True)
        vec.engine.set_option("mailbox_mode", mode)
        mb = vec.mailbox(ring=32)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            mb.run(acts_dev, 24)
            dt = (time.perf_counter() - t0) / T * 1e6
            best = dt if best is None else min(best, dt)
        out[tag + "mailbox_ring32_ahead24_dev"] = best
        mb.close()
    for k in list(out):
        if isinstance(out[k], float):
            out[k] = round(out[k], 3)
            out[k + "_env_steps_per_s"] = float(f"{B / (out[k] * 1e-6):.4g}")
    c = vec.counters()
    out["counters"] = c
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
