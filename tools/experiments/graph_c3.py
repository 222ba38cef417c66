import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pushworld_amd import benchmark_data as bd
from pushworld_amd.puzzle import PushWorldPuzzle
from pushworld_amd.vec_env import VecPushWorld
B = 65536
pool = [PushWorldPuzzle(p) for p in bd.level_paths(1)]
ids = (np.arange(B) * len(pool)) // B
vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3, border_width=1, observation="uint8", autoreset=True)
vec.reset()
K = 50
acts = torch.randint(0, 4, (K, B), dtype=torch.uint8, device=vec.device)
def loop():
    for k in range(K):
        vec.step(acts[k])
def timed(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n / K * 1e3
for rep in range(3):
    e = timed(loop, 4)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loop()
    gr = timed(g.replay, 4)
    print("eager %.4f ms/step   graph %.4f ms/step" % (e, gr))
