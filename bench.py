#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched PushWorld step engine on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c4] [--obs uint8|float32|none]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

``--gpus N`` without a torchrun environment re-executes itself through ``torch.distributed.run`` with N
ranks (one process per GPU; RCCL over xGMI carries NO data-path traffic, only: one probe all_reduce at start-up, two
barriers per timed window and, once after the last window, all_reduce(SUM) of the device counter vector, all_reduce(MAX) of
the window times and two small all_gathers of per-rank report rows); it exits non-zero when fewer than N devices are
visible, or when the probe all_reduce does not see N ranks, instead of quietly measuring fewer.

Workloads (``BASELINE.json`` / SURVEY 8d), per GPU:
  c3 (default, the headline)  65 536 environments over the 68 Level-1 puzzles (sorted by file name,
      environments grouped by puzzle), observation frame padded to the Level-1 maximum (51 x 42 cells), RGB
      render ON every step with pixels_per_cell 3 / border_width 1 / uint8, max_steps 200 with next-step
      autoreset, uniform random actions pre-generated on the device.
  c4  65 536 environments per rank of the 524 288-environment full mix: 50 % Level 0 (the 14 000 train
      puzzles of the 7 families) / 50 % Levels 1-4 (223 puzzles), N_pad 32, frame 54 x 47, action seed
      100 + rank; state only by default, ``--obs uint8`` adds the ppc-3 render.
One "step" = one pass of the hot path over the whole batch: the step kernel (dynamics, goal, reward, done)
+ the render kernel (observation).

Timing: ``--windows`` windows of EXACTLY K steps each, every window bracketed by barrier + synchronize on both
sides, per window the MAX over ranks; ``ms_per_step`` / ``value`` are the MEDIAN window (min / max reported too:
one 14 ms window is inside the +-3 % spread between observation-buffer allocations).

One JSON line of at most 4 KB is printed by rank 0 (contract in the task statement; tools/bench_line.py: top-level contract
fields + ``config`` + ``roofline`` + ``cpu_baseline`` + one short object per other configuration); the FULL record (every
sample, window, tuner candidate, sample description) goes to ``gpurun_out/bench_full.json``.  Two extra objects: ``roofline`` for
the dominant kernel (render; HBM bound), timed live by HIP events the library records around that launch on
the launch stream (``PW_OPT_PROFILE_RENDER``), and ``cpu_baseline`` = the C restatement of the reference
algorithm (oracle/pw_oracle.c, kind "port") timed on this host's cores on a bounded sample (pinned OpenMP threads, best
of three samples, all three reported), with 1 thread and with all threads, plus the pure-Python restatement of the
reference environment (the reference's own Python env cannot travel to the GPU box) on 1 core and on P processes.
``counters`` is the vector the step kernels keep on the device (pw_counters), summed over ranks by the job's
all_reduce(SUM).  ``configs`` (N = 1, c3 line; ``--no-configs`` drops it) puts every other BASELINE.json configuration on the
same clock (tools/config_suite.py): C1, C2, C3 float32 ppc 3 / ppc 20, C4 state-only + uint8, C5 on three puzzles, each
with its dominant kernel's launch time, roofline fraction, PMC traffic record and its own CPU baseline.  Non-headline
extras (``--no-extras`` drops them): ``incremental_render`` (the persistent observation buffer maintained by
pw_step_render_delta) and ``state_only_rollout`` (64 steps per launch).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
C4_SEED = 100


def level1_paths():
    d = os.path.join(ROOT, "pushworld_amd", "data", "puzzles", "level1")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".pwp")]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:  # noqa: BLE001 -- no git on the box, not a checkout
        return None


def device_report(device_index):
    """Name and clocks of the GPU the line was measured on (rocm-smi, best effort): what the render kernel reaches
    differs from box to box, so the line says which box state it saw."""
    rep = {}
    try:
        props = torch.cuda.get_device_properties(device_index)
        rep["name"] = props.name
        rep["total_memory_gb"] = round(props.total_memory / 2**30, 1)
        rep["compute_units"] = props.multi_processor_count
    except Exception:  # noqa: BLE001
        pass
    try:
        txt = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        data = json.loads(txt[txt.index("{"):])
        card = next(iter(data.values()))
        rep["rocm_smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("clock", "power", "performance"))}
    except Exception:  # noqa: BLE001 -- no rocm-smi on the box, other output format
        pass
    return rep


# ------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline(texts, ids_full, max_steps, render, pad_h, pad_w, ppc, bw, target_seconds=3.0):
    """The oracle's C port on the host cores (tools/cpu_baselines.py: same puzzle mix / action distribution / render
    settings, bounded sample, pinned OpenMP threads, best of three samples -- all three values are in the line)."""
    from tools import cpu_baselines as cb

    out = cb.port_rollout_rate(texts, ids_full, max_steps, 1 if render else 0, pad_h, pad_w, ppc, bw, seconds=target_seconds)
    out["host_cpus"] = os.cpu_count()
    out["cpu_model"] = cb.cpu_model()
    return out


def python_env_baseline(texts, ids_full, max_steps, render, pad_h, pad_w, ppc, bw, target_seconds=3.0):
    """The pure-Python restatement of the reference environment on 1 core and on P worker processes."""
    from tools import cpu_baselines as cb

    rng = np.random.default_rng(777)
    used = np.unique(np.asarray(ids_full))
    picks = [int(p) for p in rng.choice(used, size=min(4, len(used)), replace=False)]
    return cb.python_env_rate([texts[p] for p in picks], max_steps, render, pad_h, pad_w, ppc, bw, seconds=target_seconds,
                              processes=min(os.cpu_count() or 1, 128))


# ------------------------------------------------------------------------------------------ workloads
def build_workload(args, rank, world, device_index):
    """Returns a dict: vec (VecPushWorld), texts (puzzle index -> text, for the CPU baselines), ids (this rank's
    puzzle id per environment), label, frame."""
    from pushworld_amd import _capi
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.sharding import c4_global_puzzle_ids, shard_puzzle_ids
    from pushworld_amd.vec_env import VecPushWorld

    B = args.envs_per_gpu
    obs_mode = None if args.obs == "none" else args.obs
    if args.config == "c3":
        paths = level1_paths()
        texts = []
        for p in paths:
            with open(p) as f:
                texts.append(f.read())
        ids = (np.arange(B, dtype=np.int64) * len(paths)) // B  # grouped by puzzle, ~964 envs each
        vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=args.max_steps,
                           border_width=args.bw, pixels_per_cell=args.ppc, observation=obs_mode, device=device_index,
                           autoreset=True, tune_allocations=args.tune_allocations)
        label = "C3: Level-1 mix (68 puzzles, envs grouped by puzzle), step + RGB render every step" if obs_mode else \
            "C3 puzzles (Level-1 mix), state only"
        return dict(vec=vec, texts=texts, ids=ids, label=label, n_puzzles=len(texts))
    # c4: the rank's contiguous shard of the global 524 288-environment assignment, sorted inside the shard
    l0 = bd.level0_texts()  # 7 families x 2 000 train puzzles
    texts = list(l0.values())
    n_l0 = len(texts)
    for lv in (1, 2, 3, 4):
        for p in bd.level_paths(lv):
            with open(p) as f:
                texts.append(f.read())
    n_hi = len(texts) - n_l0
    total = B * world  # 524 288 at the specified 8 x 65 536
    glob = c4_global_puzzle_ids(total, n_l0, n_hi, C4_SEED)
    ids = np.sort(shard_puzzle_ids(glob, rank, world))
    assert len(ids) == B
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], device_index)
    vec = VecPushWorld(pset, B, puzzle_ids=ids, max_steps=args.max_steps, border_width=args.bw,
                       pixels_per_cell=args.ppc, observation=obs_mode, pad_cells=(54, 47), device=device_index,
                       autoreset=True, tune_allocations=args.tune_allocations)
    label = (f"C4 shard: {n_l0} Level-0 train puzzles (50 % of envs) + {n_hi} Level-1..4 puzzles (50 %), frame 54x47, "
             + ("step + uint8 ppc-3 render" if obs_mode else "state only"))
    return dict(vec=vec, texts=texts, ids=ids, label=label, n_puzzles=len(texts))


# ------------------------------------------------------------------------------------------ launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """``python bench.py --gpus N`` outside torchrun: N ranks through torch.distributed.run, one per device."""
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus and not (args.shared_device and n_dev >= 1):
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} HIP device(s) visible; refusing to measure fewer "
              f"ranks than asked (use --shared-device only to test the plumbing on one GPU)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    # the host driver of these boxes only supports dmabuf IPC: without it RCCL's cross-process buffer exchange fails with
    # "hipIpcGetMemHandle: invalid argument" (exported by the image already; kept for environments built by hand)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PUSHWORLD_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=0,
                    help="timed windows of --steps steps each (median reported); 0 = as many as fill --min-seconds, at least 7")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="with --windows 0: total timed GPU work (long enough for a device-utilisation sampler to see it)")
    ap.add_argument("--config", choices=["c3", "c4"], default="c3")
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--ppc", type=int, default=3)
    ap.add_argument("--bw", type=int, default=1)
    ap.add_argument("--obs", choices=["uint8", "float32", "none"], default=None,
                    help="default: uint8 for c3, none (state only) for c4")
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=2.0,
                    help="CPU work of ONE all-threads C-port sample (three are taken, the best is the value; the other CPU "
                         "samples scale with it)")
    ap.add_argument("--no-extras", action="store_true", help="skip the incremental-render / rollout extras")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` object (every other BASELINE.json configuration on the clock; N = 1, c3 line only)")
    ap.add_argument("--configs-only", default=None,
                    help="comma-separated prefixes of the configurations to run (C1,C2,C3_f32_ppc3,C3_f32_ppc20,C4_state,C4_u8,C5)")
    ap.add_argument("--shared-device", action="store_true",
                    help="plumbing test on a 1-GPU box: all ranks on cuda:0, gloo for the counter reduction "
                         "(RCCL refuses two ranks on one device)")
    ap.add_argument("--fused", type=int, default=0, help="1: single fused step+render launch (engine option)")
    ap.add_argument("--no-numa-pin", action="store_true", help="do not pin the launching thread to the GPU's NUMA node")
    ap.add_argument("--tune-allocations", type=int, default=None,
                    help="at most this many candidate allocations of the library-owned observation buffer (pw_obs_alloc_tuned "
                         "keeps the first one of the fast class); default: the product default of VecPushWorld")
    ap.add_argument("--full-record", default=None,
                    help="where the full record goes (default gpurun_out/bench_full.json, bench_full_n<N>.json for N > 1)")
    args = ap.parse_args()
    if args.obs is None:
        args.obs = "uint8" if args.config == "c3" else "none"
    if args.gpus < 1 or args.steps < 1 or args.windows < 0:
        raise SystemExit("--gpus and --steps must be >= 1, --windows >= 0")

    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not in_torchrun and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    n_dev = torch.cuda.device_count()
    if args.shared_device:
        device_index = 0
    else:
        if local_rank >= n_dev:
            raise SystemExit(f"bench.py: rank {rank} (local {local_rank}) has no device: {n_dev} visible, --gpus {args.gpus}")
        device_index = local_rank
    torch.cuda.set_device(device_index)
    dist = None
    backend = None
    probe_ranks = None
    # BENCH_FORCE_DIST=1: take the collective path even with one rank (smoke test of the N > 1 plumbing)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist  # RCCL over xGMI; used only for the timing barrier / reductions

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if args.shared_device else "nccl"
        try:
            if backend == "gloo":
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
                probe = torch.ones(1, device=torch.device("cuda", device_index))
                dist.all_reduce(probe)  # the first collective creates the RCCL communicator: fail here, loudly
                torch.cuda.synchronize()
                probe_ranks = int(probe.item())
                if probe_ranks != world:
                    raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
        except Exception as exc:  # noqa: BLE001 -- never fall back to fewer ranks or another backend
            print(f"bench.py: rank {rank}: torch.distributed ({backend}) failed: {exc!r}", file=sys.stderr, flush=True)
            raise SystemExit(3)
    red_dev = None if backend == "gloo" else torch.device("cuda", device_index)
    # the launching thread next to its GPU (NUMA); the CPU baseline later gets the original mask back
    from pushworld_amd.sharding import pin_to_device_numa
    numa_node, affinity_before = (None, None) if args.no_numa_pin else pin_to_device_numa(device_index)

    t_ctor = time.perf_counter()
    wl = build_workload(args, rank, world, device_index)
    constructor_s = time.perf_counter() - t_ctor  # parse + pack + tables + (observation runs) the allocator screen and the tuner
    vec = wl["vec"]
    eng = vec.engine
    dev = vec.device
    B = args.envs_per_gpu
    K, Wm = args.steps, args.warmup
    obs_mode = None if args.obs == "none" else args.obs
    if args.fused:
        eng.set_option("fused_step_render", 1)
    gen = torch.Generator(device=dev)
    gen.manual_seed((C4_SEED if args.config == "c4" else 1) + rank)
    n_act = min(Wm + K * max(args.windows, 200), 2048)  # the action stream is reused cyclically beyond that
    actions = torch.randint(0, 4, (n_act, B), generator=gen, device=dev, dtype=torch.uint8)

    vec.reset()
    vec.counters_reset()  # the device-side throughput counters (pw_counters) count the timed windows only

    step_events = []

    def one_step(t, timed=False):
        """One pass of the hot path over the batch."""
        a = actions[t % n_act]
        if obs_mode is None:
            if timed:  # the dominant kernel IS the step kernel; torch's current stream is the launch stream
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            eng.step(vec.puzzle_id, a, vec.pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated,
                     vec.flags)
            if timed:
                e1.record()
                step_events.append((e0, e1))
        else:
            # ONE call: step kernel + render launch on the same stream; with PW_OPT_PROFILE_RENDER the library
            # brackets the render launch with HIP events on that stream
            eng.step_render(vec.puzzle_id, a, vec.pos, vec.steps, vec.reward, vec.dgoals, vec.terminated,
                            vec.truncated, vec._obs_storage, vec.flags)

    def barrier():
        if dist is not None:
            dist.barrier()

    from pushworld_amd.sharding import gather_floats, gather_vectors, reduce_counters, reduce_max

    for t in range(Wm):
        one_step(t)
    M = args.windows
    if M == 0:
        # one untimed calibration window (MAX over ranks, so every rank derives the same window count)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for t in range(K):
            one_step(Wm + t)
        torch.cuda.synchronize()
        t_window = reduce_max([time.perf_counter() - t0], device=red_dev)[0]
        M = int(min(max(7, np.ceil(args.min_seconds / max(t_window, 1e-6))), max(7, 60000 // K)))
    if obs_mode is not None:
        eng.profile_render(K * M)

    torch.cuda.synchronize()
    vec.counters_reset()
    windows = []
    t_next = Wm
    for _ in range(M):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(K):
            one_step(t_next + t, timed=True)
        torch.cuda.synchronize()
        windows.append(time.perf_counter() - t0)
        barrier()
        torch.cuda.synchronize()
        t_next += K

    # the collectives of the whole job: SUM of the step counters, MAX of every window (a few bytes), and the
    # per-rank medians gathered for the report
    own_median = float(np.median(windows))
    # THE collective of the job: SUM over ranks of the counters the step kernels kept on the device
    dev_counters = vec.counters()
    counters, _ = reduce_counters(dict(dev_counters, ranks=1), own_median, device=red_dev)
    win_max = reduce_max(windows, device=red_dev)  # per window: the slowest rank
    per_rank = gather_floats(own_median, device=red_dev)
    if counters["ranks"] != world:
        raise SystemExit(f"bench.py: only {counters['ranks']} of {world} ranks reported")
    if counters["env_steps"] != B * K * M * world or dev_counters["env_steps"] != B * K * M:
        raise SystemExit(f"bench.py: the device counted {counters['env_steps']} env-steps, the loop issued {B * K * M * world}")
    # every rank's dominant-kernel time and what its allocator found (a few floats per rank)
    if obs_mode is not None:
        own_ms = np.array(eng.profile_read(), dtype=np.float64)
        assert len(own_ms) == K * M, (len(own_ms), K, M)
        tuned_ms = float(vec.tuned_ms or 0.0)
        own_gbs = B * eng.obs_bytes / (tuned_ms * 1e-3) / 1e9 if tuned_ms > 0 else 0.0
        rank_row = [float(own_ms.mean()), float(np.median(own_ms)), float(own_ms.min()), tuned_ms,
                    float(len(vec.tuned_candidates_ms)), 1.0 if own_gbs >= eng.get_option("obs_accept_gbs") else 0.0,
                    -1.0 if numa_node is None else float(numa_node), float(eng.get_option("obs_screen_ms")) * 1e-3, constructor_s]
    else:
        own_ms = np.array([a.elapsed_time(b) for a, b in step_events], dtype=np.float64)
        rank_row = [float(own_ms.mean()), float(np.median(own_ms)), float(own_ms.min()), 0.0, 0.0, 0.0,
                    -1.0 if numa_node is None else float(numa_node), 0.0, constructor_s]
    rank_rows = gather_vectors(rank_row, device=red_dev)
    elapsed = float(np.median(win_max))
    total_steps = counters["env_steps"] // M  # per window, all ranks (device-counted)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    if affinity_before is not None:  # the CPU baselines below use every core this process may use
        try:
            os.sched_setaffinity(0, affinity_before)
        except OSError:
            pass

    device_info = device_report(device_index)  # right after the timed windows: the clocks the GPU was left at
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
            "workload": wl["label"],
            "config": args.config,
            "envs_per_gpu": B,
            "global_batch": B * world,
            "puzzles": wl["n_puzzles"],
            "frame_cells": [eng.obs_shape[0] // args.ppc, eng.obs_shape[1] // args.ppc],
            "pixels_per_cell": args.ppc,
            "border_width": args.bw,
            "observation": args.obs,
            "obs_shape": list(eng.obs_shape),
            "max_steps": args.max_steps,
            "autoreset": "next-step",
            "n_pad": n_obj,
            "parallelism": f"env-sharded x{world}, no data-path collective"
                           + (f" (counters over {backend})" if backend else ""),
            # how many ranks RCCL's start-up all_reduce of ones summed to (None: no process group, N = 1)
            "ranks_in_probe_all_reduce": probe_ranks,
            # launch configuration of the page-ordered render kernel on rank 0 (pw_engine_tune_render at the first
            # reset: same bytes, the fastest of 16 page orders / occupancies for THIS observation buffer)
            "render_launch": {"tuned_index": vec.tuned_config, "tuned_ms": vec.tuned_ms,
                              "allocations_tried": len(vec.tuned_candidates_ms),
                              # wall clock of VecPushWorld(...) on rank 0 and, inside it, of the allocator's candidate screen
                              # (bounded by PW_OPT_OBS_TUNE_MS); every rank's pair is in per_rank
                              "constructor_s": round(constructor_s, 3),
                              "screen_s": round(eng.get_option("obs_screen_ms") * 1e-3, 3),
                              "screen_budget_s": round(eng.get_option("obs_tune_ms") * 1e-3, 3),
                              "allocations_max": args.tune_allocations if args.tune_allocations is not None else "product default (<= 32, within a third of the device memory)",
                              "allocator": ("pw_obs_alloc_tuned (HIP virtual-memory chunks; losers released to the device)"
                                            if getattr(vec, "obs_owned_by_library", False) else "torch (caller-owned buffer, tuned in place)"),
                              "torch_reserved_bytes": int(torch.cuda.memory_reserved(dev)),
                              "page_load_all": eng.get_option("page_load_all"),
                              "candidates_ms": [round(x, 4) for x in vec.tuned_candidates_ms],
                              "page_order": eng.get_option("page_order"),
                              "page_run_log2": eng.get_option("page_run_log2"),
                              "page_lds_pad_kb": eng.get_option("page_lds_pad_kb")},
        },
        "device": device_info,
        "timing": {
            "windows": M,
            "steps_per_window": K,
            "statistic": "median over windows of (max over ranks)",
            "window_ms_per_step": [1000.0 * w / K for w in (win_max if M <= 16 else win_max[:8] + win_max[-8:])],
            "min_ms_per_step": 1000.0 * min(win_max) / K,
            "max_ms_per_step": 1000.0 * max(win_max) / K,
            "per_rank_median_ms_per_step": [1000.0 * w / K for w in per_rank],
            # the same job read rank by rank: every rank's own median rate, summed -- next to the headline (per window the
            # SLOWEST rank) it shows whether a gap to N x the single-GPU value is one slow rank or all of them
            "sum_of_per_rank_median_rates": float(sum(B * K / w for w in per_rank)),
        },
        "counters": {k: int(v) for k, v in counters.items() if k != "ranks"},
        "counters_source": "pw_counters: kept on the device by the step kernels (one atomic per wavefront), summed over ranks by "
                           "the job's one all_reduce; env_steps == envs x steps x windows x ranks is asserted",
    }
    out["timing"]["numa_node_per_rank"] = [None if r[6] < 0 else int(r[6]) for r in rank_rows]
    if obs_mode is not None:
        ms = own_ms
        render_s = float(ms.mean()) * 1e-3
        # render launch: observation write + positions and puzzle id read
        algo = B * (eng.obs_bytes + 2 * n_obj + 4)
        achieved = algo / render_s / 1e9
        traffic, traffic_source, traffic_from = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_render_latest.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    rec = json.load(f)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from make_pmc_record import kernel_source_sha
                if rec.get("envs") == B and rec.get("obs_bytes") == eng.obs_bytes and \
                        rec.get("kernel") == eng.render_kernel and not args.fused:
                    variant = rec.get("by_page_load_all", {}).get(str(eng.get_option("page_load_all")))
                    if rec.get("kernel_source_sha16") == kernel_source_sha() and variant:
                        traffic = variant.get("hbm_bytes_per_launch")
                        traffic_from = "recorded:" + str(rec.get("kernel_source_sha16"))
                        traffic_source = "recorded: profiles/pmc_render_latest.json (rocprofv3 --pmc passes of " \
                                         + str(rec.get("source", "an earlier run")) + ", head " + str(rec.get("git_head")) \
                                         + ", same kernel source), not measured in this run"
                    else:  # a record of an older kernel describes nothing
                        traffic_source = "profiles/pmc_render_latest.json is stale (measured on another version of " \
                                         "pw_render_kernels.inc); re-run tools/collect_profiles.sh"
            except Exception:  # noqa: BLE001
                traffic = None
        kname = eng.render_kernel
        if args.fused:
            kname = ("pw_render_u8_ppc3_kernel" if kname != "pw_render_generic_kernel" else kname) + " (fused step + render)"
        out["roofline"] = {
            "kernel": kname,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_measured_copy_6290": achieved / 6290.0,
            # the MEASURED fraction: recorded PMC traffic of this launch / its duration here / the peak
            "hbm_frac": (traffic / render_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "traffic_ratio": (traffic / algo) if traffic else None,
            "traffic": traffic,
            "traffic_source": traffic_source,
            # where `traffic` comes from, in a word: "recorded:<sha16 of the kernel source the PMC passes ran on>" -- a number of
            # profiles/, not of this run (the PMC passes need rocprofv3 around the process)
            "traffic_from": traffic_from,
            "algorithmic_bytes_per_launch": algo,
            "avg_launch_ms": float(ms.mean()),
            "median_launch_ms": float(np.median(ms)),
            "min_launch_ms": float(ms.min()),
            "launches_timed": int(len(ms)),
            "timer": "HIP events recorded by the library around the launch, on the launch stream (rank 0)",
            "rest_of_step_ms": 1000.0 * elapsed / K - float(ms.mean()),  # step kernel + launch gaps
            # the same kernel on every rank (its own HIP events): fraction of the 8 TB/s peak per GPU
            "per_rank_avg_launch_ms": [r[0] for r in rank_rows],
            "per_rank_frac": [algo / (r[0] * 1e-3) / 1e9 / HBM_PEAK_GBS for r in rank_rows],
        }
        out["config"]["render_launch"]["per_rank"] = [
            {"tuned_ms": round(r[3], 4), "allocations_tried": int(r[4]), "fast_class": bool(r[5]), "screen_s": round(r[7], 3),
             "constructor_s": round(r[8], 3)} for r in rank_rows]
        out["config"]["render_launch"]["fast_class"] = bool(rank_rows[0][5])
        slow = [i for i, r in enumerate(rank_rows) if not r[5]]
        if slow:
            out["config"]["render_launch"]["ranks_without_a_fast_buffer"] = slow
        out["config"]["algorithmic_bytes_per_env_step"] = eng.obs_bytes + state_bytes
    else:
        ms = np.array([a.elapsed_time(b) for a, b in step_events], dtype=np.float64)
        algo = B * state_bytes
        achieved = algo / (float(ms.mean()) * 1e-3) / 1e9
        out["roofline"] = {
            "kernel": "pw_step_group_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
            "algorithmic_bytes_per_launch": algo, "avg_launch_ms": float(ms.mean()),
            "median_launch_ms": float(np.median(ms)), "min_launch_ms": float(ms.min()), "launches_timed": int(len(ms)),
            "timer": "HIP events on the launch stream (torch current stream), rank 0",
            "note": "state-only steps are latency / issue bound, far below the HBM roofline by construction",
            "per_rank_avg_launch_ms": [r[0] for r in rank_rows],
            "per_rank_frac": [algo / (r[0] * 1e-3) / 1e9 / HBM_PEAK_GBS for r in rank_rows],
        }
        out["config"]["algorithmic_bytes_per_env_step"] = state_bytes

    # Scaling efficiency against the N = 1 run of the same workload ON THIS HOST WITH THIS KERNEL SOURCE: value / (N x the
    # N = 1 value).  An N = 1 run leaves its value in gpurun_out/bench_n1_latest.json (untracked scratch; the driver runs
    # N = 1, 2, 4, 8 back to back from one directory); a record of other kernels, another host or another day is refused,
    # not labelled.
    from tools.config_suite import csrc_sha
    sig = {"config": args.config, "obs": args.obs, "envs_per_gpu": B, "ppc": args.ppc, "bw": args.bw, "max_steps": args.max_steps,
           "csrc_sha16": csrc_sha()}
    n1_path = os.path.join(ROOT, "gpurun_out", "bench_n1_latest.json")
    if world == 1 and not args.shared_device:
        try:
            os.makedirs(os.path.dirname(n1_path), exist_ok=True)
            with open(n1_path, "w") as f:
                json.dump({"signature": sig, "value": out["value"], "ms_per_step": out["ms_per_step"], "unix_time": time.time(),
                           "host": socket.gethostname(), "git_head": git_head()}, f, indent=1)
        except OSError:
            pass
    else:
        try:
            with open(n1_path) as f:
                rec = json.load(f)
            fresh = rec.get("host") == socket.gethostname() and time.time() - rec.get("unix_time", 0) < 6 * 3600
            if rec.get("signature") == sig and rec.get("value", 0) > 0 and fresh:
                out["scaling_efficiency"] = {
                    "value": out["value"] / (world * rec["value"]), "n1_value": rec["value"],
                    "n1_source": "gpurun_out/bench_n1_latest.json: N = 1 run on this host %d s ago, same kernel source" % (time.time() - rec["unix_time"]),
                    "sum_of_per_rank_rates_over_n_x_n1": out["timing"]["sum_of_per_rank_median_rates"] / (world * rec["value"]),
                }
            else:
                out["scaling_efficiency"] = {"value": None, "why": "the N = 1 record is stale (other kernel source, workload, host, or older than 6 h): refused"}
        except (OSError, ValueError):
            out["scaling_efficiency"] = {"value": None, "why": "no N = 1 run of this workload on this host yet (run bench.py --gpus 1 first)"}

    if not args.no_extras:
        # extra (not the headline): the same loop with the observation buffer maintained incrementally
        # (pw_step_render_delta: same bytes in HBM after every step, only the changed pixel rows written)
        if obs_mode is not None:
            try:
                def delta_step(t):
                    eng.step_render_delta(vec.puzzle_id, actions[t % n_act], vec.pos, vec.steps, vec.reward, vec.dgoals,
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
                    "note": "pw_step_render_delta on rank 0: observation buffer bit-identical to the full render "
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
        # rank 0's host cores, N = 1 only (the contract: the CPU figure belongs to the single-GPU line)
        fh, fw = eng.obs_shape[0] // args.ppc, eng.obs_shape[1] // args.ppc
        torch.cuda.synchronize()
        time.sleep(2.0)  # the host has been feeding launches: let it settle before the CPU samples
        try:
            out["cpu_baseline"] = cpu_baseline(wl["texts"], wl["ids"], args.max_steps, obs_mode is not None, fh, fw,
                                               args.ppc, args.bw, target_seconds=args.cpu_seconds)
            out["cpu_baseline"]["python_env"] = python_env_baseline(
                wl["texts"], wl["ids"], args.max_steps, obs_mode is not None, fh, fw, args.ppc, args.bw,
                target_seconds=max(0.5, 0.3 * args.cpu_seconds))
        except Exception as exc:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(exc)}
    # every other BASELINE.json configuration on the same clock (N = 1, default line only): C1, C2, C3 float32 ppc 3 / ppc 20,
    # C4 state-only + uint8, C5 on three puzzles -- each with its kernel's roofline fraction and its own CPU baseline
    if world == 1 and args.config == "c3" and not args.no_configs and not args.shared_device:
        del vec, eng
        wl.clear()
        torch.cuda.empty_cache()
        t_cfg = time.perf_counter()
        # in a CHILD process (round 6): the headline above is measured; nothing that goes wrong in one of the nine other configurations
        # -- a device fault ends the process it happens in -- may cost the line.  One retry, then the error is what the line reports.
        import subprocess

        cmd = [sys.executable, os.path.join(ROOT, "tools", "config_suite.py")]
        if args.configs_only:
            cmd += ["--only", args.configs_only]
        if args.no_cpu_baseline:
            cmd += ["--no-cpu"]
        out["configs"] = None
        for attempt in (1, 2):
            try:
                child = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
                for ln in child.stderr.splitlines():
                    if ln.startswith("config "):
                        print("bench.py: " + ln, file=sys.stderr, flush=True)
                if child.returncode == 0:
                    out["configs"] = json.loads(child.stdout.strip().splitlines()[-1])
                    break
                err = f"tools/config_suite.py ended with code {child.returncode}: " + " | ".join(child.stderr.strip().splitlines()[-3:])[:400]
            except Exception as exc:  # noqa: BLE001
                err = repr(exc)
            print(f"bench.py: configs, attempt {attempt}: {err}", file=sys.stderr, flush=True)
            out["configs"] = {"error": err, "attempts": attempt}
        out["configs"]["C3_u8_ppc3"] = {"see": "the top-level fields of this line (the headline)", "value": out["value"],
                                        "unit": out["unit"], "kernel": out["roofline"]["kernel"], "frac": out["roofline"]["frac"],
                                        "avg_launch_ms": out["roofline"]["avg_launch_ms"], "traffic": out["roofline"]["traffic"],
                                        "algorithmic_bytes_per_unit": out["roofline"]["algorithmic_bytes_per_launch"] // B}
        out["configs_wall_s"] = time.perf_counter() - t_cfg
    # the FULL record to a side file (and nothing of it to stdout: the driver keeps only the tail of stdout), the compact line
    # of at most 4 KB to stdout
    from tools.bench_line import MAX_LINE_BYTES, compact_line, dumps
    full_path = args.full_record or os.path.join("gpurun_out", "bench_full.json" if world == 1 else f"bench_full_n{world}.json")
    try:
        os.makedirs(os.path.dirname(os.path.join(ROOT, full_path)), exist_ok=True)
        with open(os.path.join(ROOT, full_path), "w") as f:
            json.dump(out, f, indent=1)
        print(f"bench.py: full record in {full_path}", file=sys.stderr, flush=True)
    except OSError as exc:
        print(f"bench.py: could not write {full_path}: {exc}", file=sys.stderr, flush=True)
        full_path = None
    line = dumps(compact_line(out, full_path))
    assert len(line) <= MAX_LINE_BYTES, len(line)
    print(line, flush=True)


if __name__ == "__main__":
    main()
