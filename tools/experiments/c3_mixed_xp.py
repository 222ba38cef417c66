import os, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from pushworld_amd.vec_env import VecPushWorld
from pushworld_amd.puzzle import PushWorldPuzzle
from pushworld_amd.config import BENCHMARK_PUZZLES_PATH
import glob
files = sorted(glob.glob(os.path.join(BENCHMARK_PUZZLES_PATH, "level1", "*.pwp")))
pool = [PushWorldPuzzle(f) for f in files]
B = 65536
ids = np.sort(np.arange(B) % len(pool))
for mode in ("never", "auto", "never", "auto"):
    env = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True, tune=False, engine_options={"step_mixed_groups": mode})
    env.reset()
    dev = env.device
    T = 64
    acts = torch.randint(0, 4, (T, B), dtype=torch.uint8, device=dev)
    env.rollout(acts); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): env.rollout(acts)
    e1.record(); e1.synchronize()
    r = 8 * T * B / (e0.elapsed_time(e1) * 1e-3)
    # single steps
    a1 = acts[0]
    for _ in range(50): env.step(a1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): env.step(a1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2000
    print(f"step_mixed_groups {mode:6s}: rollouts {r:.3e} env-steps/s   single step {dt * 1e6:.2f} us", flush=True)
