#!/usr/bin/env python3
"""Third probe: how many write fronts?  P parts, XCD k interleaving the parts k, k + 8, ... (xcd_probe_parts), pure
writes, on buffers of known class.  ms for the whole 3.79 GB buffer."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "bin", "libxcdprobe.so"))
lib.xcd_probe_parts.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                ctypes.POINTER(ctypes.c_float)]


def parts(ptr, n_pages, P, pad, reps=6):
    ms = ctypes.c_float()
    assert lib.xcd_probe_parts(ctypes.c_void_p(ptr), n_pages, P, reps, pad, ctypes.byref(ms)) == 0
    return ms.value


B = 65536
paths = bench.level1_paths()
ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                   border_width=1, observation="uint8", autoreset=True, tune=False)
vec.reset()
eng = vec.engine
cands = []
for k in range(10):
    storage, view = eng.alloc_obs(B)
    idx = eng.tune_render(vec.puzzle_id, vec.pos, storage)
    cands.append((storage, eng.get_option("tuned_ns") * 1e-6, idx))
cands.sort(key=lambda c: c[1])
torch.cuda.synchronize()
print("candidates (tuned render ms):", [round(c[1], 4) for c in cands])
PS = (8, 16, 24, 32, 40, 48, 64, 96, 128, 256, 512, 1024, 4096)
print("%-34s" % "parts:" + "".join("%8d" % P for P in PS))
for storage, ms, idx in (cands[0], cands[-1], cands[-2], cands[len(cands) // 2]):
    n_pages = storage.numel() * storage.element_size() // 4096
    for pad in (0, 4096, 7168):
        row = [parts(storage.data_ptr(), n_pages, P, pad) for P in PS]
        print("%-34s" % ("render %.4f ms, LDS pad %d:" % (ms, pad)) + "".join("%8.4f" % v for v in row), flush=True)
