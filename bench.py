#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched PushWorld step engine on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json config C3, per GPU: 65 536 environments over the 68 Level-1
puzzles (sorted by file name, environments grouped by puzzle), observation frame padded to
the Level-1 maximum (51 x 42 cells), RGB render ON every step with pixels_per_cell = 3 /
border_width = 1 / uint8, max_steps = 200 with next-step autoreset, uniform random actions
pre-generated on the device.  One "step" = one pass of the hot path over the whole batch:
the step kernel (dynamics, goal, reward, done) + the render kernel (observation).

One JSON line is printed by rank 0 (contract in the task statement) with two extra objects:
``roofline`` for the dominant kernel (render; HBM bound) measured live with HIP events on the
launch stream, and ``cpu_baseline`` = the C restatement of the reference algorithm
(oracle/pw_oracle.c, kind "port") timed on this host's cores on a bounded sample; its ``python_env``
member is the pure-Python restatement of the reference environment on one core (the reference's own
Python env cannot travel to the GPU box).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def level1_paths():
    d = os.path.join(ROOT, "pushworld_amd", "data", "puzzles", "level1")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".pwp")]


def cpu_baseline(paths, ids_full, max_steps, pad_h, pad_w, ppc, bw, target_seconds=12.0):
    """The oracle's C port on the host cores, same puzzle mix / action distribution / render
    settings, bounded sample (about 10-20 s of CPU work)."""
    from oracle import c_oracle

    puzzles = []
    for p in paths:
        with open(p) as f:
            puzzles.append(c_oracle.COraclePuzzle(f.read()))
    B = 4096
    stride = len(ids_full) // B
    ids = np.asarray(ids_full[::stride][:B], dtype=np.int32)
    rng = np.random.default_rng(12345)
    T = 16
    acts = rng.integers(0, 4, size=(T, B), dtype=np.uint8)
    c_oracle.rollout(puzzles, ids, acts, max_steps, True, pad_h, pad_w, ppc, bw)  # warm (threads, caches)
    t0 = time.perf_counter()
    _, threads = c_oracle.rollout(puzzles, ids, acts, max_steps, True, pad_h, pad_w, ppc, bw)
    dt = time.perf_counter() - t0
    rate = B * T / dt
    T2 = int(max(8, min(16384, target_seconds * rate / B)))
    acts = rng.integers(0, 4, size=(T2, B), dtype=np.uint8)
    t0 = time.perf_counter()
    c_oracle.rollout(puzzles, ids, acts, max_steps, True, pad_h, pad_w, ppc, bw)
    dt = time.perf_counter() - t0
    return {
        "value": B * T2 / dt,
        "unit": "env-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{B} envs (same Level-1 mix) x {T2} steps, step + padded uint8 render ppc={ppc}, "
                  f"OpenMP over envs, {dt:.1f} s",
        "host_cpus": os.cpu_count(),
    }


def python_env_baseline(paths, ids_full, max_steps, pad_h, pad_w, ppc, bw, target_seconds=4.0):
    """The pure-Python restatement of the reference environment (oracle/pw_oracle.py: hash-set collision
    tables, per-cell painter, /255 + np.pad -- the closest thing to the reference's own CPU Python env that
    can travel to this box), one process, same puzzle mix, step + padded observation."""
    from oracle import pw_oracle

    rng = np.random.default_rng(777)
    picks = rng.choice(len(paths), size=4, replace=False)
    envs = []
    t_build0 = time.perf_counter()
    for p in picks:
        with open(paths[p]) as f:
            envs.append(pw_oracle.OracleEnv(pw_oracle.OraclePuzzle(f.read()), max_steps))
    t_build = time.perf_counter() - t_build0
    for e in envs:
        e.reset()
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < target_seconds:
        for e in envs:
            state, _, term, trunc = e.step(int(rng.integers(0, 4)))
            e.puzzle.observation_u8(state, pad_h, pad_w, ppc, bw)
            if term or trunc:
                e.reset()
            steps += 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port (pure Python)",
            "sample": f"4 Level-1 puzzles x {steps // 4} steps, step + padded uint8 render ppc={ppc}, {dt:.1f} s; "
                      f"collision-table construction of the 4 puzzles took {t_build:.1f} s (not included)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--ppc", type=int, default=3)
    ap.add_argument("--bw", type=int, default=1)
    ap.add_argument("--obs", choices=["uint8", "float32", "none"], default="uint8")
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused", type=int, default=0,
                    help="0: pw_step then pw_render (events bracket the render kernel); 1: one pw_step_render call")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dist = None
    # BENCH_FORCE_DIST=1: take the RCCL path even with one rank (smoke test of the N > 1 plumbing on a 1-GPU box)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist  # RCCL over xGMI; used only for the timing barrier/reduction

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = level1_paths()
    B = args.envs_per_gpu
    K, Wm = args.steps, args.warmup
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B  # grouped by puzzle, ~964 envs each
    obs_mode = None if args.obs == "none" else args.obs
    vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=args.max_steps,
                       border_width=args.bw, pixels_per_cell=args.ppc, observation=obs_mode,
                       device=local_rank, autoreset=True)
    eng = vec.engine
    dev = vec.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    actions = torch.randint(0, 4, (K + Wm, B), generator=gen, device=dev, dtype=torch.uint8)

    vec.reset()

    def one_step(t, events=None):
        """One pass of the hot path over the batch.  The HIP events bracket the dominant kernel only
        (the render launch; with --fused 1 the single fused launch)."""
        a = actions[t]
        if obs_mode is None:
            if events is not None:
                events[0].record()
            eng.step(vec.puzzle_id, a, vec.pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated,
                     vec.flags)
        elif args.fused:
            # ONE launch: wave 0 of each workgroup advances its environment, the workgroup draws it
            if events is not None:
                events[0].record()
            eng.step_render(vec.puzzle_id, a, vec.pos, vec.steps, vec.reward, vec.dgoals, vec.terminated,
                            vec.truncated, vec._obs_storage, vec.flags)
        else:
            # step kernel (tens of microseconds) + render kernel on the same stream
            eng.step(vec.puzzle_id, a, vec.pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated,
                     vec.flags)
            if events is not None:
                events[0].record()
            eng.render(vec.puzzle_id, vec.pos, vec._obs_storage)
        if events is not None:
            events[1].record()

    for t in range(Wm):
        one_step(t)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        one_step(Wm + t, evs[t])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # the ONLY collective of the whole job: SUM of counters, MAX of the window (a few bytes)
    from pushworld_amd.sharding import reduce_counters

    counters, elapsed = reduce_counters({"env_steps": B * K}, elapsed, device=dev)
    total_steps = counters["env_steps"]

    if rank == 0:
        n_obj = eng.np
        state_bytes = 2 * n_obj * 2 + 1 + 4 + 4 * 2 + 8 + 1 + 1 + 1  # pos r/w, action, pid, steps r/w, reward, flags
        out = {
            "metric": "env-steps/sec",
            "value": total_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": 1000.0 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if args.obs != "float32" else "f32",
            "data": "synthetic",
            "config": {
                "workload": "C3: Level-1 mix (68 puzzles, envs grouped by puzzle), step + RGB render every step",
                "envs_per_gpu": B,
                "global_batch": B * world,
                "frame_cells": [eng.obs_shape[0] // args.ppc, eng.obs_shape[1] // args.ppc],
                "pixels_per_cell": args.ppc,
                "border_width": args.bw,
                "observation": args.obs,
                "obs_shape": list(eng.obs_shape),
                "max_steps": args.max_steps,
                "autoreset": "next-step",
                "n_pad": n_obj,
                "parallelism": f"env-sharded x{world}, no data-path collective",
            },
        }
        if obs_mode is not None:
            ms = np.array([a.elapsed_time(b) for a, b in evs])
            render_s = float(ms.mean()) * 1e-3
            # render launch: observation write + positions and puzzle id read;
            # the fused launch also carries the step's state traffic
            algo = B * (eng.obs_bytes + (state_bytes if args.fused else 2 * n_obj + 4))
            achieved = algo / render_s / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_render_latest.json")
            if os.path.exists(pmc):
                try:
                    with open(pmc) as f:
                        rec = json.load(f)
                    if rec.get("envs") == B and rec.get("obs_bytes") == eng.obs_bytes and \
                            rec.get("kernel") == eng.render_kernel and not args.fused:
                        traffic = rec.get("hbm_bytes_per_launch")
                except Exception:  # noqa: BLE001
                    traffic = None
            out["roofline"] = {
                "kernel": ("pw_render_u8_ppc3_kernel (fused step + render)" if (args.fused and eng.render_kernel != "pw_render_generic_kernel")
                           else eng.render_kernel + (" (fused step + render)" if args.fused else "")),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_copy_6290": achieved / 6290.0,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": algo,
                "avg_launch_ms": float(ms.mean()),
                "min_launch_ms": float(ms.min()),
                "rest_of_step_ms": 1000.0 * elapsed / K - float(ms.mean()),  # step kernel + launch gaps
            }
            out["config"]["algorithmic_bytes_per_env_step"] = eng.obs_bytes + state_bytes
        # extra (not the headline): the same loop with the observation buffer maintained incrementally
        # (pw_step_render_delta: same bytes in HBM after every step, only the changed pixel rows written)
        if obs_mode is not None:
            try:
                def delta_step(t):
                    eng.step_render_delta(vec.puzzle_id, actions[t], vec.pos, vec.steps, vec.reward, vec.dgoals,
                                          vec.terminated, vec.truncated, vec._obs_storage, vec.flags)
                for t in range(Wm):
                    delta_step(t)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for t in range(K):
                    delta_step(Wm + t)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                out["incremental_render"] = {
                    "env_steps_per_s": B * K / dt, "ms_per_step": 1000.0 * dt / K, "n_gpus": 1,
                    "note": "pw_step_render_delta on this rank: observation buffer bit-identical to the full render "
                            "after every step (tests/test_gpu_incremental.py); not the headline value",
                }
            except Exception as exc:  # noqa: BLE001
                out["incremental_render"] = {"error": repr(exc)}
        # extra (not the headline): the same batch state-only, T steps per launch (pw_rollout)
        try:
            Tn = 64
            racts = torch.randint(0, 4, (Tn, B), generator=gen, device=dev, dtype=torch.uint8)
            vec.rollout(racts)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                vec.rollout(racts)
            e1.record()
            torch.cuda.synchronize()
            out["state_only_rollout"] = {"env_steps_per_s": 4 * Tn * B / (e0.elapsed_time(e1) * 1e-3),
                                          "steps_per_launch": Tn, "envs": B, "n_gpus": 1}
        except Exception as exc:  # noqa: BLE001
            out["state_only_rollout"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(paths, ids, args.max_steps, eng.obs_shape[0] // args.ppc,
                                                   eng.obs_shape[1] // args.ppc, args.ppc, args.bw)
                out["cpu_baseline"]["python_env"] = python_env_baseline(
                    paths, ids, args.max_steps, eng.obs_shape[0] // args.ppc, eng.obs_shape[1] // args.ppc, args.ppc,
                    args.bw)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
