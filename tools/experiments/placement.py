#!/usr/bin/env python3
"""Does the render time depend on where the observation buffer sits?  Same process, same engine, the
buffer taken at different offsets of one big slab and from fresh allocations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pushworld_amd import _capi  # noqa: E402
from pushworld_amd import benchmark_data as bd  # noqa: E402
from pushworld_amd.puzzle import PushWorldPuzzle  # noqa: E402
from pushworld_amd.vec_env import VecPushWorld  # noqa: E402

B = 65536
pool = [PushWorldPuzzle(p) for p in bd.level_paths(1)]
ids = (np.arange(B) * len(pool)) // B
vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3, border_width=1, observation="uint8", autoreset=True)
vec.reset()
eng = vec.engine
stride = eng.obs_stride
nbytes = B * stride
dev = vec.device


def time_render(storage, n=30):
    for _ in range(3):
        _capi.check(_capi.lib.pw_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos), _capi._ptr(storage), stride, B, eng._stream()))
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _capi.check(_capi.lib.pw_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos), _capi._ptr(storage), stride, B, eng._stream()))
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    # proxy: plain fill of the same bytes
    evs = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        storage.view(-1)[:nbytes].zero_()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    f = np.array([a.elapsed_time(b) for a, b in evs])
    global LAST_FILL
    LAST_FILL = float(np.median(f))
    print("      fill median %.4f" % LAST_FILL, end="  ")
    _capi.check(_capi.lib.pw_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos), _capi._ptr(storage), stride, B, eng._stream()))
    return float(np.median(t)), float(t.min())


def al(ptr):
    k = 0
    while k < 40 and ptr % (1 << (k + 1)) == 0:
        k += 1
    return "ptr=0x%x align=2^%d" % (ptr, k)


print("engine's own buffer: %s  median %.4f min %.4f" % ((al(vec._obs_storage.data_ptr()),) + time_render(vec._obs_storage)))
slab = torch.empty((nbytes + (64 << 20),), dtype=torch.uint8, device=dev)
base = slab.data_ptr()
print("slab base", al(base))
for off in (0, 4096, 2 << 20):
    v = slab[off:off + nbytes]
    print("slab offset %9d: median %.4f min %.4f" % ((off,) + time_render(v)))
keep = []
for i in range(6):
    t = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    keep.append(t)
    print("fresh allocation %d: %s median %.4f min %.4f" % ((i, al(t.data_ptr())) + time_render(t)))
# again the first ones (drift over time?)
print("engine's own buffer again: median %.4f min %.4f" % time_render(vec._obs_storage))
print("slab offset 0 again: median %.4f min %.4f" % time_render(slab[0:nbytes]))

del keep
torch.cuda.empty_cache()
# sizes: does a larger allocation (other size class / alignment) behave differently?
for extra in (0, 2 << 20, 64 << 20, 1 << 30):
    t = torch.empty((nbytes + extra,), dtype=torch.uint8, device=dev)
    print("alloc nbytes+%d: %s median %.4f min %.4f" % ((extra, al(t.data_ptr())) + time_render(t[:nbytes])))
    del t
    torch.cuda.empty_cache()
# zeros vs empty, 2-D vs 1-D
t = torch.zeros((B, stride), dtype=torch.uint8, device=dev)
print("zeros 2-D: %s median %.4f min %.4f" % ((al(t.data_ptr()),) + time_render(t)))
del t
torch.cuda.empty_cache()
print("engine's own buffer again: median %.4f min %.4f" % time_render(vec._obs_storage))
# everything else freed: a new VecPushWorld-sized buffer FIRST in a clean cache
