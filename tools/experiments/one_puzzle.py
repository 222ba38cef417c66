import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pushworld_amd import _capi
from pushworld_amd import benchmark_data as bd
from pushworld_amd.vec_env import VecPushWorld
B = 65536
name = sys.argv[1]
paths = [p for lv in (1, 2, 3, 4) for p in bd.level_paths(lv)]
i = [k for k, p in enumerate(paths) if name in p][0]
only = len(sys.argv) > 2
pset = _capi.PuzzleSet([_capi.ParsedPuzzle(open(p).read()) for p in (paths[i:i+1] if only else paths)], 0)
vec = VecPushWorld(pset, B, puzzle_ids=np.full(B, 0 if only else i), max_steps=200, observation=None, autoreset=True)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 4, (64, B), generator=g, device="cuda", dtype=torch.uint8)
nterm = 0
for k in range(128):
    _, r, t, u = vec.step(acts[k % 64])
    nterm += int(t.sum())
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for k in range(128):
    vec.step(acts[k % 64])
b.record()
torch.cuda.synchronize()
print(name, "np", vec.engine.np, "us/step", a.elapsed_time(b) / 128 * 1e3, "terminations per step per env", nterm / 128 / B)
