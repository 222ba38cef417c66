#!/usr/bin/env python3
"""(HISTORICAL: written against the round-4 adapter, whose scalars lived in a device buffer; since round 5 the product IS
variant 2 below and `_raw_host` no longer exists -- kept for the record of what was measured.)
Round-4 experiment, AS RUN at commit e7ae6e8 (before the product adopted its result -- pushworld_amd/_single_env.py now keeps
the observation in pinned host memory itself, so "device buffer + copy" below no longer is what `PushWorldEnv` does): the
single-environment adapters with the observation buffer in pinned HOST memory (the render kernels write it over PCIe, no
copy command) against the device buffer + hipMemcpy: 9.7 k -> 18.1 k steps/s at max_steps 50; the state there too: 18.2 k."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pushworld_amd import _single_env
from pushworld_amd.gym_env import PushWorldEnv
from pushworld_amd.config import BENCHMARK_PUZZLES_PATH

path = os.path.join(BENCHMARK_PUZZLES_PATH, "level0", "base", "train", "level_0_base_train_0.pwp")
if not os.path.exists(path):
    import glob
    path = sorted(glob.glob(os.path.join(BENCHMARK_PUZZLES_PATH, "level1", "*.pwp")))[0]


def run(env, n, check=None):
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 4, size=n)
    env.reset(seed=1)
    obs_all = []
    t0 = time.perf_counter()
    for a in acts:
        o, r, term, trunc, info = env.step(int(a))
        if check is not None:
            obs_all.append((o.copy(), r, term, trunc))
        if term or trunc:
            o, _ = env.reset()
            if check is not None:
                obs_all.append((o.copy(), 0.0, False, False))
    return n / (time.perf_counter() - t0), obs_all


base = PushWorldEnv(path, max_steps=50)
rate0, ref = run(base, 300, check=True)
rate0, _ = run(base, 5000)

host = PushWorldEnv(path, max_steps=50)
core = host._core if hasattr(host, "_core") else host
eng = core._engine
esz = 4
stor = torch.zeros((1, eng.obs_stride // esz), dtype=torch.float32).pin_memory()
h, w, c = eng.obs_shape
view = stor.as_strided((1, h, w, c), (eng.obs_stride // esz, w * c, c, 1))
core._obs_storage, core._obs = stor, view


def read_back(self=core):
    self._raw_host.copy_(self._raw, non_blocking=True)
    torch.cuda.current_stream(self._engine.device).synchronize()
    return self._obs[0].numpy().copy(), self._raw_host.numpy()


core._read_back = read_back
rate1, got = run(host, 300, check=True)
same = len(got) == len(ref) and all(np.array_equal(a[0], b[0]) and a[1:] == b[1:] for a, b in zip(got, ref))
rate1, _ = run(host, 5000)
# variant 2: the per-step scalars and the positions in pinned host memory too (the step kernel reads / writes them over PCIe)
host2 = PushWorldEnv(path, max_steps=50)
core2 = host2._core if hasattr(host2, "_core") else host2
eng2 = core2._engine
stor2 = torch.zeros((1, eng2.obs_stride // esz), dtype=torch.float32).pin_memory()
view2 = stor2.as_strided((1, h, w, c), (eng2.obs_stride // esz, w * c, c, 1))
core2._obs_storage, core2._obs = stor2, view2
npad = eng2.np
raw2 = torch.zeros((16 + 2 * npad,), dtype=torch.uint8).pin_memory()
core2._raw = raw2
core2._buf = {"reward": raw2[0:8].view(torch.float64), "steps": raw2[8:12].view(torch.int32), "terminated": raw2[12:13],
              "truncated": raw2[13:14], "dgoals": raw2[14:15].view(torch.int8), "pos": raw2[16:].view(torch.int8).view(1, npad, 2)}


def read_back2(self=core2):
    torch.cuda.current_stream(self._engine.device).synchronize()
    return self._obs[0].numpy().copy(), self._raw.numpy().copy()


core2._read_back = read_back2
rate2, got2 = run(host2, 300, check=True)
same2 = len(got2) == len(ref) and all(np.array_equal(a[0], b[0]) and a[1:] == b[1:] for a, b in zip(got2, ref))
rate2, _ = run(host2, 5000)
print(f"state in host memory too: {rate2:9.0f} steps/s   {'same' if same2 else 'DIFFERENT'}")
print(f"device buffer + copy: {rate0:9.0f} steps/s   host-resident observation: {rate1:9.0f} steps/s   {'same' if same else 'DIFFERENT'}")
