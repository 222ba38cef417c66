#!/usr/bin/env python3
"""Which observation buffers are "fast class" for the page-ordered render kernel?  Same process, same engine, the C3
batch rendered into (A) ten separate allocations, (B) ten consecutive windows of one 40 GB slab, (C) windows of the
slab shifted by small offsets.  Two launch configurations per buffer: the default (runs of 64 pages + 7 KB) and
the eighths order + 7 KB (the one whose speed depends on the buffer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pushworld_amd import _capi  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402

B = 65536
paths = bench.level1_paths()
ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                   border_width=1, observation="uint8", autoreset=True, tune=False)
vec.reset()
eng = vec.engine
stride = eng.obs_stride
nbytes = B * stride


def timed(ptr_tensor, cfg, reps=12):
    for k, v in zip(("page_order", "page_run_log2", "page_lds_pad_kb"), cfg):
        eng.set_option(k, v)
    call = lambda: _capi.check(_capi.lib.pw_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos),  # noqa: E731
                                                   _capi._ptr(ptr_tensor), stride, B, eng._stream()))
    for _ in range(2):
        call()
    eng.profile_render(reps)
    for _ in range(reps):
        call()
    ms = np.array(eng.profile_read())
    eng.profile_render(0)
    return float(np.median(ms))


def row(name, t):
    print("%-34s ptr %#014x  default %.4f  eighths %.4f" % (name, t.data_ptr(), timed(t, (2, 6, 7)), timed(t, (1, 0, 7))), flush=True)


print("A: separate allocations")
keep = []
for i in range(10):
    t = torch.empty((nbytes,), dtype=torch.uint8, device=vec.device)
    keep.append(t)
    row("alloc %d" % i, t)
print("A again (same buffers, later)")
for i in (0, 3, 7):
    row("alloc %d" % i, keep[i])
del keep
torch.cuda.empty_cache()
print("B: windows of one slab")
slab = torch.empty((10 * nbytes + (64 << 20),), dtype=torch.uint8, device=vec.device)
for i in range(10):
    row("slab window %d" % i, slab[i * nbytes:(i + 1) * nbytes])
print("C: shifted windows of the slab")
for off in (4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 33 << 20):
    row("slab window 0 + %d" % off, slab[off:off + nbytes])
    row("slab window 5 + %d" % off, slab[5 * nbytes + off:6 * nbytes + off])
