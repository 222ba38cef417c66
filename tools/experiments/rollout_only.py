import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pushworld_amd import benchmark_data as bd
from pushworld_amd.puzzle import PushWorldPuzzle
from pushworld_amd.vec_env import VecPushWorld
B = 65536
pool = [PushWorldPuzzle(p) for p in bd.level_paths(1)]
ids = (np.arange(B) * len(pool)) // B
vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True)
vec.reset()
acts = torch.randint(0, 4, (64, B), dtype=torch.uint8, device=vec.device)
for _ in range(6):
    vec.rollout(acts)
torch.cuda.synchronize()
