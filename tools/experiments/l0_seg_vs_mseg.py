"""Level-0 puzzles in the segment kernel (many environments of ONE puzzle per launch) against the many-puzzle lane kernel
(64 puzzles per wavefront): microseconds per step of 64-step launches, 16 384 environments."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pushworld_amd import _capi, benchmark_data as bd
from pushworld_amd.vec_env import VecPushWorld
from tools.bench_bind import timed

texts = list(bd.level0_texts(limit=40).values())  # 280 puzzles
B = 16384
pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
acts = torch.as_tensor(np.random.default_rng(0).integers(0, 4, size=(64, B), dtype=np.uint8)).cuda()
for opts in ({"step_boards": "never"},):
    vec = VecPushWorld(pset, B, puzzle_ids=np.zeros(B, np.int64), max_steps=100, observation=None, device=0, autoreset=True, bind=True, engine_options=opts)
    rows = []
    for i in list(range(0, 280, 23)):
        vec.set_puzzle_ids(np.full(B, i, np.int64)); vec.reset(); vec.rollout(acts)
        med, _ = timed(lambda: vec.rollout(acts), 3, reps=3)
        rows.append(round(med / 64, 2))
    print("one puzzle per launch (segments):", rows, vec.bound_info)
    ids = np.sort(np.random.default_rng(1).integers(0, 280, B))
    # 280 puzzles x ~58 environments: bound at min_envs 48 -> raise it so that everything goes through the many-puzzle kernel
    vec.engine.set_option("bind_min_envs", 4096)
    vec.set_puzzle_ids(ids); vec.reset(); vec.rollout(acts)
    med, _ = timed(lambda: vec.rollout(acts), 3, reps=3)
    print("280 puzzles, many per wavefront:", round(med / 64, 2), vec.bound_info)
    vec.engine.set_option("bind_min_envs", 8)
    vec.set_puzzle_ids(ids); vec.reset(); vec.rollout(acts)
    med, _ = timed(lambda: vec.rollout(acts), 3, reps=3)
    print("280 puzzles, a segment each:", round(med / 64, 2), vec.bound_info)
