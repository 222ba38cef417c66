#!/usr/bin/env python3
"""Where the single-step kernel's time goes: ablated builds (-DPW_STEP_XP=n, tools/experiments/bin/libpw_xpN.so via
PUSHWORLD_AMD_LIB) of pw_step_group_kernel on C3 and the C4 shard, one step per launch, state only.
  0 production   1 no push set (moved = 0)   2 agent wall test only   3 empty kernel   4 puzzle-id load + one store"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pushworld_amd import _capi  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402
from tools.experiments.step_ab import timed, workloads  # noqa: E402

print("library:", os.path.basename(_capi.LIB_PATH))
for name, pool, B, ids in workloads():
    vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, observation=None, autoreset=True)
    vec.engine.set_option("step_lds_tables", 2)
    vec.reset()
    g = torch.Generator(device=vec.device).manual_seed(1)
    acts = torch.randint(0, 4, (64, B), generator=g, device=vec.device, dtype=torch.uint8)
    it = [0]

    def one():
        vec.step(acts[it[0] % 64])
        it[0] += 1

    med, mn = timed(one, 100)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300):
        one()
    b.record()
    torch.cuda.synchronize()
    print("%-28s step us med %7.2f  min %7.2f   300 back to back: %7.2f us each" % (name, med, mn, a.elapsed_time(b) / 300 * 1e3), flush=True)
    del vec
