#!/usr/bin/env python3
"""Throughput of the BASELINE.json configs that are not the bench.py headline (one GPU each):

  C2  4 096 copies of one Level-0 puzzle, state only (per-launch pw_step and pw_rollout K steps/launch)
  C4  one rank's shard of the 8-GPU config: 65 536 envs, 50 % Level 0 (7 families' train sets) +
      50 % Levels 1-4, frame 54 x 47, state only and with the uint8 ppc-3 render
  C4r the same shard run through PushWorldVectorEnv-style episode turnover (pw_resample + autoreset)

    python tools/bench_configs.py [--l0-per-family N]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def timed(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--l0-per-family", type=int, default=None, help="limit of Level-0 puzzles per family (default all 2000)")
    args = ap.parse_args()
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.vec_env import VecPushWorld

    out = {}
    dev = torch.device("cuda", 0)

    # ---------------------------------------------------------------- C1: the reference's own usage pattern
    import tempfile
    from pushworld_amd.gym_env import PushWorldEnv
    from pushworld_amd.puzzle import PushWorldPuzzle

    text = list(bd.level0_texts(("base",), "train", 1).values())[0]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "level_0_base_train_0.pwp")
        with open(path, "w") as f:
            f.write(text)
        rng = np.random.default_rng(0)
        for name, kw in (("default_ppc20_f32", {}), ("ppc3_bw1", dict(pixels_per_cell=3, border_width=1))):
            env = PushWorldEnv(path, max_steps=100, **kw)
            env.reset(seed=0)
            n = 3000
            acts = rng.integers(0, 4, n)
            for a in acts[:50]:
                env.step(int(a))
            t0 = time.perf_counter()
            for a in acts:
                _, _, term, trunc, _ = env.step(int(a))
                if term or trunc:
                    env.reset()
            dt = time.perf_counter() - t0
            out["C1_gym_step_" + name] = {"steps_per_s": n / dt, "obs_shape": list(env.observation_space.shape)}
        pz = PushWorldPuzzle(path)
        s = pz.initial_state
        t0 = time.perf_counter()
        for a in acts:
            s = pz.get_next_state(s, int(a))
        out["C1_get_next_state_batch1"] = {"steps_per_s": n / (time.perf_counter() - t0)}

    # ---------------------------------------------------------------- C2
    t0 = time.perf_counter()
    l0 = bd.load_level0(("base",), "train", 1)
    vec = VecPushWorld(l0, 4096, max_steps=100, observation=None, autoreset=True)
    vec.reset()
    T = 64
    acts = torch.randint(0, 4, (T, 4096), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    it = [0]

    def one():
        vec.step(acts[it[0] % T])
        it[0] += 1

    dt = timed(one, 2000, 50)
    out["C2_step_per_launch"] = {"ms": dt * 1e3, "env_steps_per_s": 4096 / dt}
    dt = timed(lambda: vec.rollout(acts), 200, 5)
    out["C2_rollout_64_per_launch"] = {"ms": dt * 1e3, "env_steps_per_s": 4096 * T / dt}
    # the same 64 single-step launches captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(T):
            vec.step(acts[k])
    dt = timed(g.replay, 200, 5)
    out["C2_hipgraph_64_step_launches"] = {"ms": dt * 1e3, "env_steps_per_s": 4096 * T / dt}
    vec_r = VecPushWorld(l0, 4096, max_steps=100, observation="uint8", pixels_per_cell=3, border_width=1, autoreset=True,
                         incremental=True)
    vec_r.reset()
    for k in range(4):
        vec_r.step(acts[k])

    def loop_r():
        for k in range(T):
            vec_r.step(acts[k])

    dt = timed(loop_r, 20, 2)
    out["C2_with_incremental_render_eager"] = {"ms_per_step": dt * 1e3 / T, "env_steps_per_s": 4096 * T / dt}
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        loop_r()
    dt = timed(g2.replay, 50, 3)
    out["C2_with_incremental_render_hipgraph"] = {"ms_per_step": dt * 1e3 / T, "env_steps_per_s": 4096 * T / dt}

    # ---------------------------------------------------------------- C4 shard
    t0 = time.perf_counter()
    pool = bd.load_level0(limit=args.l0_per_family) + bd.load_levels()
    n_l0 = len(pool) - 223
    t_parse = time.perf_counter() - t0
    B = 65536
    rng = np.random.default_rng(100)
    ids = np.concatenate([rng.integers(0, n_l0, B // 2), n_l0 + rng.integers(0, 223, B // 2)])
    ids.sort()
    for name, obs in (("state_only", None), ("render_u8_ppc3", "uint8")):
        t0 = time.perf_counter()
        vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, observation=obs, pixels_per_cell=3, border_width=1,
                           pad_cells=(54, 47), autoreset=True)
        t_engine = time.perf_counter() - t0
        vec.reset()
        acts = torch.randint(0, 4, (T, B), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        it[0] = 0

        def one4():
            vec.step(acts[it[0] % T])
            it[0] += 1

        dt = timed(one4, 200, 10)
        ent = {"ms": dt * 1e3, "env_steps_per_s": B / dt, "np": vec.engine.np, "puzzles": len(pool),
               "engine_create_s": t_engine}
        if obs is None:
            dt = timed(lambda: vec.rollout(acts), 20, 2)
            ent["rollout_64_ms"] = dt * 1e3
            ent["rollout_env_steps_per_s"] = B * T / dt
        else:
            ent["render_kernel"] = vec.engine.render_kernel
            ent["obs_bytes"] = vec.engine.obs_bytes
            ent["write_TBps"] = B * vec.engine.obs_bytes / dt / 1e12
        out["C4_" + name] = ent
        del vec
    out["C4_parse_all_puzzles_s"] = t_parse
    # packed pool file (f2): save once, load instead of parsing
    import tempfile
    from pushworld_amd import _capi
    t0 = time.perf_counter()
    pset = _capi.PuzzleSet([p._parsed for p in pool], 0)
    t_pack = time.perf_counter() - t0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "pool.pwset")
        t0 = time.perf_counter()
        pset.save(path)
        t_save = time.perf_counter() - t0
        t0 = time.perf_counter()
        back = _capi.PuzzleSet.load(path, 0)
        t_load = time.perf_counter() - t0
        out["C4_packed_pool_file"] = {"bytes": os.path.getsize(path), "pack_upload_s": t_pack, "save_s": t_save,
                                      "load_upload_s": t_load, "puzzles": len(back)}
    del pset, back

    # ---------------------------------------------------------------- C4 with device-side episode turnover
    table = np.concatenate([np.repeat(np.arange(n_l0), 223), np.tile(n_l0 + np.arange(223), n_l0)]) if n_l0 * 223 * 2 < 1 << 24 \
        else None
    if table is None:  # equal total weight of the two halves with a short table: sample L0 ids
        sub = rng.integers(0, n_l0, 223 * 16)
        table = np.concatenate([sub, np.tile(n_l0 + np.arange(223), 16)])
    vec = VecPushWorld(pool, B, max_steps=200, observation="uint8", pixels_per_cell=3, border_width=1, pad_cells=(54, 47),
                       autoreset=True, resample=table, seed=100)
    vec.reset(seed=100)
    acts = torch.randint(0, 4, (T, B), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    it[0] = 0

    def one5():
        vec.step(acts[it[0] % T])
        it[0] += 1

    dt = timed(one5, 400, 250)  # past the first truncation wave: puzzles are shuffled across envs by then
    out["C4_render_resample"] = {"ms": dt * 1e3, "env_steps_per_s": B / dt}
    del vec
    vec = VecPushWorld(pool, B, max_steps=200, observation="uint8", pixels_per_cell=3, border_width=1, pad_cells=(54, 47),
                       autoreset=True, resample=table, seed=100, incremental=True)
    vec.reset(seed=100)
    it[0] = 0
    dt = timed(one5, 400, 250)
    out["C4_render_resample_incremental"] = {"ms": dt * 1e3, "env_steps_per_s": B / dt}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
