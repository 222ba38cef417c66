"""Every BASELINE.json configuration on the clock: the ``configs`` object of bench.py's JSON line.

Each entry: ``value`` / ``unit`` (whole-job throughput on ONE GPU, inputs resident in HBM), ``kernel`` (the dominant
launch), ``algorithmic_bytes_per_unit``, ``units_per_launch``, ``avg_launch_ms`` (HIP events the library records
around that launch on the launch stream: PW_OPT_PROFILE_RENDER), ``achieved_gbs`` = algorithmic bytes per launch / average
launch time, ``frac`` of the 8 TB/s HBM peak, ``traffic`` (HBM bytes per launch from committed rocprofv3 PMC passes of the
same kernel source, or null) and its own ``cpu_baseline`` (the oracle's C port on a sample of the same workload).

  C1  one Level-0 puzzle, batch 1, the gym adapter (default ppc 20 / float32) with and without the observation, the
      single-state API (get_next_state, render_plan)
  C2  4 096 copies of that puzzle, state only: one step per launch and 64-step rollouts
  C3  65 536 Level-1 environments: float32 ppc 3 (the uint8 ppc 3 headline is bench.py's main line) and float32 ppc 20 on
      8 192 environments
  C4  one rank's 65 536-environment shard of the 524 288-environment full mix: state only and with the uint8 ppc-3 render
  C5  pw_expand4 on breadth-first frontiers of `2 Obstacle`, `Pull Dont Push`, `Four Pistons` -- >= 4 M distinct states or,
      for smaller state spaces, as many disjoint buffer sets as take the footprint beyond the 256 MB Infinity Cache

    python tools/config_suite.py [--only C2,C5] [--no-cpu]        (prints the object; bench.py embeds it)
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0
# CPU baselines of the non-headline configurations: short bounded samples (the whole default bench.py run has to fit a minute;
# the headline's own cpu_baseline takes three samples of --cpu-seconds each)
CPU_SECONDS, CPU_SAMPLES = 0.6, 2
CSRC = os.path.join(ROOT, "pushworld_amd", "csrc")


def csrc_sha():
    """sha256 over the kernel sources: a PMC record describes the code it was measured on and nothing else."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".inc", ".hip", ".h", ".cpp")):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


# which kernel sources a configuration's PMC record describes (its dominant kernel's translation-unit pieces): a record stays
# valid while THESE files are unchanged -- a tweak to the search kernels does not invalidate the render records
_RENDER_SRC = ("pw_render_kernels.inc", "pw_zone.h", "pw_format.h")
_STEP_SRC = ("pw_step_kernels.inc", "pw_seg_kernels.inc", "pw_format.h")  # (bound batches: the segment role lives in pw_seg_kernels.inc)
_EXPAND_SRC = ("pw_expand_kernels.inc", "pw_format.h")  # (pw_expand4_v2*_kernel: self-contained in that file)
KEY_SOURCES = {"C3_f32_ppc3": _RENDER_SRC, "C3_f32_ppc20_8192": _RENDER_SRC, "C4_u8_ppc3": _RENDER_SRC,
               "C4_state": _STEP_SRC, "C4_rollout": _STEP_SRC, "C2_step": _STEP_SRC, "C2_rollout": _STEP_SRC,
               "C5_2_obstacle": _EXPAND_SRC, "C5_pull_dont_push": _EXPAND_SRC, "C5_four_pistons": _EXPAND_SRC}


def key_sha(config_key):
    """sha256 over the source files the record of ``config_key`` depends on."""
    h = hashlib.sha256()
    for name in KEY_SOURCES.get(config_key, ()):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(config_key, units_per_launch):
    """(traffic bytes per launch, source string) from profiles/pmc_kernels_latest.json when it was measured on this
    kernel source and launch size; (None, why) otherwise."""
    path = os.path.join(ROOT, "profiles", "pmc_kernels_latest.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/pmc_kernels_latest.json"
    ent = rec.get("configs", {}).get(config_key)
    if not ent:
        return None, "profiles/pmc_kernels_latest.json has no record for " + config_key
    current = ent.get("source_sha16") == key_sha(config_key) if "source_sha16" in ent else rec.get("csrc_sha16") == csrc_sha()
    if not current:
        return None, ("the record of " + config_key + " in profiles/pmc_kernels_latest.json is stale (measured on another version of "
                      + ", ".join(KEY_SOURCES.get(config_key, ("pushworld_amd/csrc",))) + "); re-run tools/collect_profiles.sh")
    if int(ent.get("units_per_launch", -1)) != int(units_per_launch):
        return None, f"record is for {ent.get('units_per_launch')} units per launch, this run has {units_per_launch}"
    return ent["hbm_bytes_per_launch"], (f"recorded: profiles/pmc_kernels_latest.json ({ent.get('kernel_symbol')}; rocprofv3 --pmc FETCH_SIZE / "
                                          f"WRITE_SIZE passes of {rec.get('source')}, head {rec.get('git_head')}, same kernel source), "
                                          "not measured in this run")


def wall(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def launch_ms(eng, fn, launches):
    """Durations (ms) of ``launches`` dominant-kernel launches issued by ``fn()``, from the library's HIP events."""
    eng.profile_render(launches)
    for _ in range(launches):
        fn()
    ms = np.array(eng.profile_read(), dtype=np.float64)
    eng.profile_render(0)
    return ms


def entry(value, unit, workload, kernel, bytes_per_unit, units_per_launch, ms, config_key=None, **extra):
    """``frac`` is the MODEL fraction: SURVEY 8d's algorithmic bytes of the launch / its average duration / the 8 TB/s peak.
    ``hbm_frac`` is the MEASURED one: the recorded PMC traffic of the same launch / the same duration / the peak;
    ``traffic_ratio`` = traffic / algorithmic bytes (1.0 = the launch moves exactly what the model charges)."""
    avg = float(ms.mean())
    achieved = bytes_per_unit * units_per_launch / (avg * 1e-3) / 1e9
    out = {"value": value, "unit": unit, "workload": workload, "kernel": kernel,
           "algorithmic_bytes_per_unit": bytes_per_unit, "units_per_launch": int(units_per_launch),
           "avg_launch_ms": avg, "median_launch_ms": float(np.median(ms)), "min_launch_ms": float(ms.min()),
           "launches_timed": int(len(ms)), "achieved_gbs": achieved, "peak_gbs": HBM_PEAK_GBS, "frac": achieved / HBM_PEAK_GBS,
           "frac_is": "model: algorithmic bytes per launch / average launch time / peak",
           "timer": "HIP events recorded by the library around the launch, on the launch stream"}
    if config_key:
        out["traffic"], out["traffic_source"] = pmc_traffic(config_key, units_per_launch)
        if out["traffic"]:
            out["hbm_frac"] = out["traffic"] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["traffic_ratio"] = out["traffic"] / (bytes_per_unit * units_per_launch)
    out.update(extra)
    return out


def state_bytes(npad):
    # positions read + written, action, puzzle id, step counter r/w, flags, dgoals, f64 reward
    return 4 * npad + 22


def rollout_bytes(npad, T):
    """Algorithmic bytes per ENV-STEP of a T-step pw_rollout launch: the state is read once and written once per launch
    (positions r/w, puzzle id, step counter r/w, the last step's reward / delta / flags), one action byte per step."""
    return (4 * npad + 21 + T) / T


def actions_for(T, B, dev, seed):
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    return torch.randint(0, 4, (T, B), generator=gen, device=dev, dtype=torch.uint8)


# ------------------------------------------------------------------------------------------------ C1
def run_c1(cpu=True):
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.gym_env import PushWorldEnv
    from pushworld_amd.puzzle import PushWorldPuzzle

    member, text = next(iter(bd.level0_texts(("base",), "train", 1).items()))
    out = {"workload": f"C1: {member} (7 x 7 with its border, 3 movables), batch 1, actions numpy default_rng(0).integers(0, 4)"}
    rng = np.random.default_rng(0)
    n = 10000
    acts = rng.integers(0, 4, n)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, os.path.basename(member))
        with open(path, "w") as f:
            f.write(text)
        env = PushWorldEnv(path, max_steps=100)  # the reference's defaults: ppc 20, border 2, float32 observation
        env.reset(seed=0)
        for a in acts[:100]:
            env.step(int(a))
        env.reset(seed=0)
        m = 1500
        rates = []  # three segments: a host-side latency loop -- the boxes' CPUs are shared, one busy neighbour halves a segment
        for seg in range(3):
            t0 = time.perf_counter()
            for a in acts[seg * m:(seg + 1) * m]:
                _, _, term, trunc, _ = env.step(int(a))
                if term or trunc:
                    env.reset()
            rates.append(m / (time.perf_counter() - t0))
        dt = m / max(rates)
        eng = env._engine
        b = env._buf
        ms = launch_ms(eng, lambda: eng.render(env._pid, b["pos"], env._obs_storage), 200)
        out["gym_step_with_render"] = entry(
            m / dt, "env-steps/s", "gym PushWorldEnv.step (+ reset on done), default ppc 20 / border 2 / float32 observation returned as a host array",
            eng.render_kernel, int(eng.obs_bytes), 1, ms, obs_shape=list(env.observation_space.shape), samples=rates,
            sample="best of 3 segments of 1 500 steps (all three in `samples`)",
            note="batch 1 is launch / copy latency, not bandwidth: the fraction is reported because every entry has one")
        pz = PushWorldPuzzle(path)
        s = pz.initial_state
        for a in acts[:200]:
            s = pz.get_next_state(s, int(a))
        s = pz.initial_state
        t0 = time.perf_counter()
        for a in acts:
            s = pz.get_next_state(s, int(a))
            if pz.is_goal_state(s):
                s = pz.initial_state
        dt = time.perf_counter() - t0
        out["get_next_state_batch1"] = {
            "value": n / dt, "unit": "calls/s", "kernel": "pw_plan_kernel (one wavefront; state in the kernel arguments, result in mapped pinned host memory)",
            "us_per_call": 1e6 * dt / n, "algorithmic_bytes_per_unit": state_bytes(4),
            "note": "PushWorldPuzzle.get_next_state (puzzle.py:348-394) = pw_next_state: one launch, no copy command, completion word polled"}
        plan = [int(a) for a in acts[:100]]
        pz.render_plan(plan)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            frames = pz.render_plan(plan)
        out["render_plan_100_steps"] = {"value": 1e3 * (time.perf_counter() - t0) / reps, "unit": "ms", "frames": len(frames),
                                        "frame_shape": list(frames[0].shape),
                                        "note": "one pw_plan_states launch + ONE batched pw_render + one copy to the host (puzzle.py:471-506)"}
        t0 = time.perf_counter()
        for _ in range(20):
            pz.is_valid_plan(plan)
        out["is_valid_plan_100_steps"] = {"value": 1e3 * (time.perf_counter() - t0) / 20, "unit": "ms"}
        out["value"], out["unit"] = out["gym_step_with_render"]["value"], "env-steps/s"
        if cpu:
            from tools.cpu_baselines import python_env_rate, port_rollout_rate

            w, h = pz.dimensions
            # (the float32 / 255 observation the gym path returns -- like for like with gym_step_with_render)
            py_r = python_env_rate([text], 100, True, h, w, 20, 2, seconds=1.0, obs_dtype="f32")
            py_s = python_env_rate([text], 100, False, h, w, 20, 2, seconds=0.6)
            cp = port_rollout_rate([text], np.zeros(64, np.int64), 100, 0, h, w, 20, 2, seconds=0.4, samples=CPU_SAMPLES, sample_envs=64, with_one_thread=False)
            out["cpu_baseline"] = {"value": py_r["value"], "unit": "env-steps/s", "cores": 1, "kind": "port (pure Python)",
                                   "sample": py_r["sample"] + " [float32 observation ppc 20, as gym_env.py:188-226 returns it]",
                                   "state_only": {"value": py_s["value"], "sample": py_s["sample"]},
                                   "c_port_state_only_all_threads": {"value": cp["value"], "cores": cp["cores"], "sample": cp["sample"]}}
    return out


# ------------------------------------------------------------------------------------------------ C2
def run_c2(cpu=True):
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    member, text = next(iter(bd.level0_texts(("base",), "train", 1).items()))
    B, T = 4096, 64
    vec = VecPushWorld([PushWorldPuzzle(text=text)], B, max_steps=100, observation=None, autoreset=True)
    vec.reset()
    dev = vec.device
    acts = actions_for(T, B, dev, 0)
    it = [0]

    def one():
        vec.step(acts[it[0] % T])
        it[0] += 1

    dt = wall(one, 4000, 100)
    ms = launch_ms(vec.engine, one, 500)
    # (4 096 copies of one puzzle: every environment sits in a segment of the bound batch -- the segments, not the whole-grid boards)
    all_bound = bool(vec.bound_info) and vec.bound_info["bound_envs"] == B
    kern = "pw_step_seg_kernel" if all_bound else ("pw_step_board_kernel" if vec.engine.get_option("step_board_set") else "pw_step_group_kernel")
    sb = state_bytes(vec.engine.np)
    out = entry(B / dt, "env-steps/s", f"C2: 4 096 copies of {member}, state only, max_steps 100, next-step autoreset, one step per launch",
                kern, sb, B, ms, "C2_step", ms_per_step=1e3 * dt,
                note="64 wavefronts on the whole chip: launch / latency bound by construction (SURVEY 8d)")
    dtr = wall(lambda: vec.rollout(acts), 300, 5)
    msr = launch_ms(vec.engine, lambda: vec.rollout(acts), 100)
    out["rollout_64_steps_per_launch"] = entry(B * T / dtr, "env-steps/s", "the same batch, pw_rollout: 64 steps per launch (state in registers "
                                               "between the steps: per launch the state once in, once out + 64 action bytes)", kern,
                                               rollout_bytes(vec.engine.np, T), B * T, msr, "C2_rollout")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(T):
            vec.step(acts[k])
    dtg = wall(g.replay, 200, 5)
    out["hipgraph_64_single_step_launches"] = {"value": B * T / dtg, "unit": "env-steps/s", "us_per_step": 1e6 * dtg / T}
    # the resident kernel (pw_mailbox_*): what a step costs a host that waits for its verdicts -- next to pw_step + synchronise
    import time as _time

    def sync_steps(n):
        for k in range(n):
            vec.step(acts[k % T])
            torch.cuda.current_stream().synchronize()

    sync_steps(50)
    torch.cuda.synchronize()
    t0 = _time.perf_counter()
    sync_steps(1000)
    us_launch_sync = (_time.perf_counter() - t0) / 1000 * 1e6
    acts_run = acts.repeat(16, 1)[:1024].contiguous()
    with vec.mailbox(ring=8) as mb:
        mb.run(acts_run, 1)
        us = {}
        for name, ahead in (("sync", 1), ("ahead8", 8)):
            best = None
            for _ in range(3):
                t0 = _time.perf_counter()
                mb.run(acts_run, ahead)
                d = (_time.perf_counter() - t0) / len(acts_run) * 1e6
                best = d if best is None else min(best, d)
            us[name] = best
    out["mailbox_step"] = {"value": B / (us["sync"] * 1e-6), "unit": "env-steps/s", "us_per_step": us["sync"],
                           "us_per_step_8_in_flight": us["ahead8"], "pw_step_plus_synchronise_us": us_launch_sync,
                           "workload": "the same batch through pw_mailbox_run: every step posted by the host through a pinned word and "
                                       "waited for (verdicts in pinned host memory); no launch, no stream synchronisation"}
    c = vec.counters()
    out["counters"] = c
    if cpu:
        from tools.cpu_baselines import port_rollout_rate

        out["cpu_baseline"] = port_rollout_rate([text], np.zeros(B, np.int64), 100, 0, 7, 7, 3, 1, seconds=CPU_SECONDS, samples=CPU_SAMPLES, with_one_thread=False)
    return out


# ------------------------------------------------------------------------------------------------ C3 variants
def run_c3(obs, ppc, bw, B, key, steps, cpu=True, cpu_envs=4096):
    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    texts = [open(p).read() for p in paths]
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    t0 = time.perf_counter()
    vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=200, border_width=bw,
                       pixels_per_cell=ppc, observation=obs, autoreset=True)
    t_build = time.perf_counter() - t0
    vec.reset()
    eng = vec.engine
    T = 32
    acts = actions_for(T, B, vec.device, 1)
    it = [0]

    def one():
        vec.step(acts[it[0] % T])
        it[0] += 1

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    time.sleep(1.0)  # the driver clears the released candidate buffers in the background for a moment
    dt = wall(one, steps, 3)
    ms = launch_ms(eng, one, steps)
    per_env = int(eng.obs_bytes) + 2 * eng.np + 4  # render launch: observation write + positions and record read
    out = entry(B / dt, "env-steps/s",
                f"C3: {B} envs over the 68 Level-1 puzzles (grouped by puzzle), frame 51 x 42, step + {obs} ppc {ppc} / border {bw} render every step",
                eng.render_kernel, per_env, B, ms, key, ms_per_step=1e3 * dt, obs_shape=list(eng.obs_shape),
                algorithmic_bytes_per_env_step=int(eng.obs_bytes) + state_bytes(eng.np), constructor_s=t_build,
                render_launch={"tuned_index": vec.tuned_config, "tuned_ms": vec.tuned_ms, "allocations_tried": len(vec.tuned_candidates_ms)})
    out["counters"] = vec.counters()
    del vec
    torch.cuda.empty_cache()
    if cpu:
        from tools.cpu_baselines import port_rollout_rate

        out["cpu_baseline"] = port_rollout_rate(texts, ids, 200, "f32" if obs == "float32" else "u8", 51, 42, ppc, bw, seconds=CPU_SECONDS, samples=CPU_SAMPLES,
                                                sample_envs=cpu_envs, with_one_thread=False)
    return out


# ------------------------------------------------------------------------------------------------ C4
def run_c4(obs, key, cpu=True):
    import bench

    B = 65536
    args = argparse.Namespace(envs_per_gpu=B, obs=obs or "none", config="c4", max_steps=200, bw=1, ppc=3, tune_allocations=None)
    t0 = time.perf_counter()
    wl = bench.build_workload(args, 0, 8, torch.cuda.current_device())
    t_build = time.perf_counter() - t0
    vec = wl["vec"]
    eng = vec.engine
    vec.reset()
    T = 64
    acts = actions_for(T, B, vec.device, 100)
    it = [0]

    def one():
        vec.step(acts[it[0] % T])
        it[0] += 1

    if obs is None:
        dt = wall(one, 2000, 50)
        ms = launch_ms(eng, one, 500)
        sb = state_bytes(eng.np)
        out = entry(B / dt, "env-steps/s", wl["label"] + " (rank 0 of 8), one step per launch through VecPushWorld.step",
                    "pw_step_group_mixed_kernel", sb, B, ms, key, ms_per_step=1e3 * dt, build_s=t_build,
                    note="latency / issue bound, far below the HBM roofline by construction (SURVEY 8d)")
        dtr = wall(lambda: vec.rollout(acts), 40, 3)
        msr = launch_ms(eng, lambda: vec.rollout(acts), 20)
        out["rollout_64_steps_per_launch"] = entry(B * T / dtr, "env-steps/s", "the same shard, pw_rollout: 64 steps per launch (state in "
                                                   "registers between the steps: per launch the state once in, once out + 64 action bytes)",
                                                   "pw_step_seg_kernel + pw_step_mseg_kernel" if vec.bound_info else "pw_step_group_mixed_kernel",
                                                   rollout_bytes(eng.np, T), B * T, msr, "C4_rollout")
    else:
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        time.sleep(1.0)
        dt = wall(one, 200, 3)
        ms = launch_ms(eng, one, 200)
        per_env = int(eng.obs_bytes) + 2 * eng.np + 4
        out = entry(B / dt, "env-steps/s", wl["label"] + " (rank 0 of 8)", eng.render_kernel, per_env, B, ms, key, ms_per_step=1e3 * dt,
                    obs_shape=list(eng.obs_shape), algorithmic_bytes_per_env_step=int(eng.obs_bytes) + state_bytes(eng.np), build_s=t_build,
                    render_launch={"tuned_index": vec.tuned_config, "tuned_ms": vec.tuned_ms, "allocations_tried": len(vec.tuned_candidates_ms)})
    out["counters"] = vec.counters()
    texts, ids = wl["texts"], wl["ids"]
    del vec, wl
    torch.cuda.empty_cache()
    if cpu:
        from tools.cpu_baselines import port_rollout_rate

        out["cpu_baseline"] = port_rollout_rate(texts, ids, 200, "u8" if obs else 0, 54, 47, 3, 1, seconds=CPU_SECONDS, samples=CPU_SAMPLES, with_one_thread=False)
    return out


# ------------------------------------------------------------------------------------------------ C5
C5_PUZZLES = (("C5_2_obstacle", "level1/2 Obstacle.pwp"), ("C5_pull_dont_push", "level2/Pull Dont Push.pwp"),
              ("C5_four_pistons", "level4/Four Pistons.pwp"))
CACHE_BUST_BYTES = 640 << 20  # the Infinity Cache is 256 MB: the buffers a timed pass cycles through are well beyond it


def c5_frontier(rel, target_states=4_000_000):
    """(puzzle, int32 [F, N] Position2D) -- the first F distinct states of a GPU breadth-first search in C++ object order."""
    from pushworld_amd.config import BENCHMARK_PUZZLES_PATH
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch

    pz = PushWorldPuzzle(os.path.join(BENCHMARK_PUZZLES_PATH, rel), order="cpp")
    bfs = BreadthFirstSearch(pz, max_states=target_states + (target_states >> 1))
    bfs.begin()
    try:
        while bfs.total_states < target_states and not bfs.exhausted:
            bfs.expand()
    except ValueError:  # the store filled up inside a layer: what is in it is enough
        pass
    F = min(bfs.total_states, target_states)
    xy = bfs.states(0, F)
    exhausted = bfs.exhausted
    bfs.close()
    st = (xy[:, :, 0].astype(np.int64) * 10000 + xy[:, :, 1]).astype(np.int32)
    distinct = len(st)
    if distinct < target_states:  # a state space smaller than the frontier asked for (`2 Obstacle`: 39 023 states): repeated
        st = np.ascontiguousarray(np.tile(st, (-(-target_states // distinct), 1))[:target_states])
    return pz, st, exhausted, distinct


def run_c5(key, rel, cpu=True):
    pz, st_host, exhausted, distinct = c5_frontier(rel)
    F, N = st_host.shape
    dev = pz._engine().device
    per_parent = 20 * N + 20
    nbuf = max(1, -(-CACHE_BUST_BYTES // (F * per_parent)))
    sets = []
    for _ in range(nbuf):
        sets.append((torch.as_tensor(st_host).to(dev), torch.empty((F, 4, N), dtype=torch.int32, device=dev),
                     torch.empty((F, 4), dtype=torch.int32, device=dev), torch.empty((F, 4), dtype=torch.uint8, device=dev)))
    eng = pz._engine()
    it = [0]

    def one():
        s = sets[it[0] % nbuf]
        it[0] += 1
        eng.expand4(0, s[0], s[1], s[2], s[3])

    reps = max(40, 4 * nbuf)
    dt = min(wall(one, reps, nbuf) for _ in range(3))  # (a timed region of a few milliseconds: one hiccup of the host doubles it)
    ms = launch_ms(eng, one, reps)
    out = entry(F / dt, "parents/s",
                f"C5: pw_expand4 on {F} states of a breadth-first search of {rel} in discovery order (C++ object order, N = {N}; {distinct} distinct"
                + (" = the whole state space, repeated; " if distinct < F else "; ") + f"{nbuf} disjoint buffer set(s), {nbuf * F * per_parent / 2**20:.0f} MB per cycle: "
                "beyond the 256 MB Infinity Cache)",
                ("pw_expand4_v2_kernel" if eng.get_option("expand_lds_tables") == 0 and 2 <= N <= 16 else "pw_expand4_lane_kernel")
                if F >= 131072 else "pw_expand4_kernel", per_parent, F, ms, key,
                ms_per_launch=1e3 * dt, movables=int(N), states=int(F), distinct_states=int(distinct), buffer_sets=int(nbuf),
                successors_per_s=4 * F / dt)
    del sets
    torch.cuda.empty_cache()
    if cpu:
        from pushworld_amd.config import BENCHMARK_PUZZLES_PATH
        from tools.cpu_baselines import port_expand_rate

        with open(os.path.join(BENCHMARK_PUZZLES_PATH, rel)) as f:
            out["cpu_baseline"] = port_expand_rate(f.read(), st_host, seconds=0.4, samples=CPU_SAMPLES)
    return out


# ------------------------------------------------------------------------------------------------ driver
def run_all(only=None, cpu=True, log=None):
    """The ``configs`` object.  A configuration that fails is reported as {"error": ...}: the others still run."""
    jobs = [
        ("C1", lambda: run_c1(cpu)),
        ("C2", lambda: run_c2(cpu)),
        ("C3_f32_ppc3", lambda: run_c3("float32", 3, 1, 65536, "C3_f32_ppc3", 100, cpu)),
        ("C3_f32_ppc20_8192", lambda: run_c3("float32", 20, 2, 8192, "C3_f32_ppc20_8192", 24, cpu, cpu_envs=256)),
        ("C4_state", lambda: run_c4(None, "C4_state", cpu)),
        ("C4_u8_ppc3", lambda: run_c4("uint8", "C4_u8_ppc3", cpu)),
    ] + [(k, (lambda k=k, rel=rel: run_c5(k, rel, cpu))) for k, rel in C5_PUZZLES]
    out = {}
    for name, fn in jobs:
        if only and not any(name.startswith(o) for o in only):
            continue
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001
            import traceback

            out[name] = {"error": repr(exc), "traceback": traceback.format_exc()[-1500:]}
        out[name]["wall_s"] = time.perf_counter() - t0
        if log:
            log(f"config {name}: {out[name].get('value', out[name].get('error'))} {out[name].get('unit', '')} "
                f"({out[name]['wall_s']:.1f} s)")
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, help="comma-separated prefixes: C1,C2,C3_f32_ppc3,C4_state,C5 ...")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    res = run_all(args.only.split(",") if args.only else None, not args.no_cpu, log=lambda s: print(s, file=sys.stderr, flush=True))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
