import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pushworld_amd import benchmark_data as bd
from pushworld_amd.puzzle import PushWorldPuzzle
from pushworld_amd.vec_env import VecPushWorld
texts = [open(p).read() for p in bd.level_paths(1)]
B, T = 6000, 40
mk = lambda bind: VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, max_steps=20, observation=None, device=0, autoreset=True, resample=True, seed=9, bind=bind)
a, b = mk(True), mk(False)
a.reset(); b.reset()
acts = torch.as_tensor(np.random.default_rng(1).integers(0, 4, size=(T, B), dtype=np.uint8)).to(a.device)
for t in range(T):
    a.step(acts[t]); b.step(acts[t])
    torch.cuda.synchronize()
    bad = (a.pos != b.pos).flatten(1).any(1).nonzero().flatten().cpu().numpy()
    sb = (a.steps != b.steps).nonzero().flatten().cpu().numpy()
    if len(bad) or len(sb):
        print("t", t, "bad pos", len(bad), bad[:10], "bad steps", len(sb), sb[:10], "mism", a.engine.get_option("bind_mismatches"))
        e = int(bad[0]) if len(bad) else int(sb[0])
        print(" env", e, "pid", int(a.puzzle_id[e]), int(b.puzzle_id[e]), "steps", int(a.steps[e]), int(b.steps[e]), "term", int(a.terminated[e]), int(b.terminated[e]), int(a.truncated[e]), int(b.truncated[e]))
        print(" a", a.pos[e].flatten().tolist()); print(" b", b.pos[e].flatten().tolist())
        ids = a.puzzle_id.cpu().numpy(); print(" count of pid", (ids == ids[e]).sum(), "bad pids", np.unique(ids[bad])[:20])
        break
print("done", a.engine.get_option("bind_mismatches"))
