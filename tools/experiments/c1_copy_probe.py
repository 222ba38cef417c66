#!/usr/bin/env python3
"""Round-6 probe: the parts of a C1 gym step (graph replay + completion word, the host copy of the observation, the rest) and what a
copy of the observation costs when its lines were just written by the device (cold) against a second copy of the same lines (warm)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pushworld_amd import benchmark_data as bd
from pushworld_amd.gym_env import PushWorldEnv

member, text = next(iter(bd.level0_texts(("base",), "train", 1).items()))
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, os.path.basename(member))
open(path, "w").write(text)
env = PushWorldEnv(path, max_steps=100)
env.reset(seed=0)
acts = np.random.default_rng(0).integers(0, 4, 6000)
for a in acts[:200]:
    _, _, te, tr, _ = env.step(int(a))
    if te or tr:
        env.reset()
pc = time.perf_counter
t0 = pc()
for a in acts[:3000]:
    _, _, te, tr, _ = env.step(int(a))
    if te or tr:
        env.reset()
print("gym step (+ resets)    %.2f us" % ((pc() - t0) / 3000 * 1e6), "obs", env._obs_np.shape, env._obs_np.nbytes, "bytes; graphs:", bool(env._graphs))
def parts(label):
    T = dict(launch=0.0, wait=0.0, cold=0.0)
    word = env._signal_np
    n = 3000
    for a in acts[:n]:
        a = int(a)
        t0 = pc()
        if env._graphs:
            word[0] = 0
            env._graphs[a].replay()
            t1 = pc()
            while word[0] == 0:
                pass
        else:
            rc = env._step_call(env._acts_ptr + a)
            env._signalled += 1
            want = env._signalled
            t1 = pc()
            while word[0] < want:
                pass
        t2 = pc()
        o = env._obs_np.copy()
        t3 = pc()
        T["launch"] += t1 - t0; T["wait"] += t2 - t1; T["cold"] += t3 - t2
    print(label, {k: round(v / n * 1e6, 2) for k, v in T.items()}, "us")

parts("as built (fused=%d, graphs=%s)" % (env._engine.get_option("step_one_fused"), bool(env._graphs)))
env._engine.set_option("step_one_fused", 0)
env._graphs = False
parts("two launches, eager")
env._graphs = None
for a in acts[:50]:
    env.step(int(a))
parts("two launches, graphs=%s" % bool(env._graphs))
