#!/usr/bin/env python3
"""Second probe: the contiguous-parts write pattern by hand on buffers of known class.  Every XCD writes its own part
of the buffer (xcd_probe_multi): all eight at once (= the eighths page order without any rendering), with the XCDs
permuted over the parts, subsets of the XCDs, pairs of parts."""
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
lib.xcd_probe_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_int,
                                ctypes.POINTER(ctypes.c_float)]


def multi(bases, pages, reps=6, pad=7168):
    b = (ctypes.c_void_p * 8)(*[ctypes.c_void_p(int(x)) for x in bases])
    n = (ctypes.c_uint32 * 8)(*[int(x) for x in pages])
    ms = ctypes.c_float()
    assert lib.xcd_probe_multi(b, n, reps, pad, ctypes.byref(ms)) == 0
    return sum(int(x) for x in pages) * 4096 / (ms.value * 1e-3) / 1e9, ms.value


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
for storage, ms, idx in (cands[0], cands[1], cands[-1], cands[-2]):
    nbytes = storage.numel() * storage.element_size()
    pe = nbytes // 8 // 4096  # pages per eighth
    base = storage.data_ptr()
    parts = [base + k * pe * 4096 for k in range(8)]
    print("\nbuffer with tuned render %.4f ms (tuner index %d)" % (ms, idx))
    for pad in (0, 7168):
        gb, t = multi(parts, [pe] * 8, pad=pad)
        print("  eighths by hand, all 8 XCDs, LDS pad %4d B: %6.0f GB/s (%.4f ms for the buffer)" % (pad, gb, t))
    rng = np.random.default_rng(0)
    for trial in range(4):
        perm = rng.permutation(8)
        gb, t = multi([parts[perm[k]] for k in range(8)], [pe] * 8)
        print("  XCD k on part %s: %6.0f GB/s" % (perm.tolist(), gb))
    for sub in ([0, 1, 2, 3], [4, 5, 6, 7], [0, 2, 4, 6], [1, 3, 5, 7], [0, 1], [0, 4], [0, 1, 2, 3, 4, 5]):
        gb, t = multi(parts, [pe if k in sub else 0 for k in range(8)])
        print("  only XCDs %-20s (own parts): %6.0f GB/s = %5.0f per XCD" % (sub, gb, gb / len(sub)))
    # quarters: XCD pairs (k, k + 4) interleave the pages of quarter k -- approximated: both write halves of the quarter
    q = nbytes // 4 // 4096
    gb, t = multi([base + (k % 4) * q * 4096 + (k // 4) * (q // 2) * 4096 for k in range(8)], [q // 2] * 8)
    print("  quarters by hand (XCD k and k + 4 on the two halves of quarter k %% 4): %6.0f GB/s" % gb)
