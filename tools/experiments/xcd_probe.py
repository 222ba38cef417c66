#!/usr/bin/env python3
"""XCD x region write-bandwidth map of observation buffers of known class (tools/experiments/xcd_probe.hip).
For a few candidate buffers: the tuner's best time (class), then GB/s of each single XCD into each 256 MB region."""
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
lib.xcd_probe.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]


def probe(ptr, nbytes, mask, reps=4):
    ms = ctypes.c_float()
    assert lib.xcd_probe(ctypes.c_void_p(ptr), nbytes, mask, reps, ctypes.byref(ms)) == 0
    return nbytes / (ms.value * 1e-3) / 1e9


B = 65536
paths = bench.level1_paths()
ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                   border_width=1, observation="uint8", autoreset=True, tune=False)
vec.reset()
eng = vec.engine
REGION = 256 << 20
cands = []
for k in range(8):
    storage, view = eng.alloc_obs(B)
    idx = eng.tune_render(vec.puzzle_id, vec.pos, storage)
    cands.append((storage, eng.get_option("tuned_ns") * 1e-6, idx))
cands.sort(key=lambda c: c[1])
torch.cuda.synchronize()
for storage, ms, idx in (cands[0], cands[-1], cands[len(cands) // 2]):
    nbytes = storage.numel() * storage.element_size()
    nreg = nbytes // REGION
    print("\nbuffer with tuned render %.4f ms (tuner index %d), %d regions of 256 MB; all 8 XCDs on the whole buffer: %.0f GB/s"
          % (ms, idx, nreg, probe(storage.data_ptr(), nreg * REGION, 0xFF)))
    print("GB/s of one XCD alone into each region (rows = XCD 0..7):")
    m = np.zeros((8, nreg))
    for x in range(8):
        for r in range(nreg):
            m[x, r] = probe(storage.data_ptr() + r * REGION, REGION, 1 << x)
        print("  " + " ".join("%5.0f" % v for v in m[x]))
    print("  all 8 XCDs into each region: " + " ".join("%5.0f" % probe(storage.data_ptr() + r * REGION, REGION, 0xFF) for r in range(nreg)))
    # the eighths pattern by hand: XCD k alone on eighth k, then all eight at once
    e = (nbytes // 8) // 4096 * 4096
    print("  XCD k alone on eighth k:     " + " ".join("%5.0f" % probe(storage.data_ptr() + k * e, e, 1 << k) for k in range(8)))
    print("  XCD k alone on eighth k+1:   " + " ".join("%5.0f" % probe(storage.data_ptr() + ((k + 1) % 8) * e, e, 1 << k) for k in range(8)))
