#!/usr/bin/env python3
"""The C3 render into buffers whose PHYSICAL layout is chosen with the HIP virtual-memory API
(tools/experiments/vmm_alloc.hip): one chunk, chunks of 2 MB .. 512 MB in creation order, in a random permutation,
interleaved.  Does the "allocation class" of the contiguous-parts page orders follow the physical layout?"""
import ctypes
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

vmm = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "bin", "libvmm.so"))
vmm.vmm_alloc.restype = ctypes.c_int
vmm.vmm_alloc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64,
                          ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
vmm.vmm_free.argtypes = [ctypes.c_void_p]

B = 65536
paths = bench.level1_paths()
ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                   border_width=1, observation="uint8", autoreset=True, tune=False)
vec.reset()
eng = vec.engine
stride = eng.obs_stride
nbytes = B * stride
CFGS = [("default", (2, 6, 7)), ("eighths+7K", (1, 0, 7)), ("quarters+7K", (1, 1, 7)), ("quarters+5K", (1, 1, 5))]


def timed(ptr, cfg, reps=10):
    for k, v in zip(("page_order", "page_run_log2", "page_lds_pad_kb"), cfg):
        eng.set_option(k, v)
    call = lambda: _capi.check(_capi.lib.pw_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos),  # noqa: E731
                                                   ctypes.c_void_p(ptr), stride, B, eng._stream()))
    for _ in range(2):
        call()
    eng.profile_render(reps)
    for _ in range(reps):
        call()
    ms = np.array(eng.profile_read())
    eng.profile_render(0)
    return float(np.median(ms))


print("%-44s" % "buffer" + "".join("%14s" % n for n, _ in CFGS))
ref = vec._obs_storage
print("%-44s" % "torch allocation (engine's)" + "".join("%14.4f" % timed(ref.data_ptr(), c) for _, c in CFGS), flush=True)
MB = 1 << 20
CASES = [(0, 0, 0, "vmm: one chunk")] * 2
for mb in (2, 64, 256, 512):
    CASES += [(1, mb * MB, 0, "vmm: %d MB chunks in order" % mb)] * 2 + [(4, mb * MB, 0, "vmm: %d MB chunks reversed" % mb)] * 2
CASES += [(2, 64 * MB, 1, "vmm: 64 MB chunks permuted"), (3, 64 * MB, 0, "vmm: 64 MB, evens then odds")]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if sys.argv[1] in c[3]]
for mode, chunk, seed, label in CASES:
    p, h = ctypes.c_void_p(), ctypes.c_void_p()
    n = vmm.vmm_alloc(0, nbytes, chunk, mode, seed, ctypes.byref(p), ctypes.byref(h))
    if n <= 0:
        print("%-44s failed (%d)" % (label, n))
        continue
    row = "".join("%14.4f" % timed(p.value, c) for _, c in CFGS)
    # correctness of the last render into this buffer
    got = torch.empty_like(ref)
    _capi.check(_capi.lib.pw_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos), _capi._ptr(ref), stride, B, eng._stream()))
    torch.cuda.synchronize()
    import ctypes as _c
    hip = _c.CDLL("libamdhip64.so")
    hip.hipMemcpy(_c.c_void_p(got.data_ptr()), p, _c.c_size_t(nbytes), 3)
    ok = torch.equal(got, ref)
    print("%-44s" % ("%s [%d chunks]%s" % (label, n, "" if ok else " MISMATCH")) + row, flush=True)
    # not freed: unmapping and re-reserving the range faults on this runtime (the buffers are 3.8 GB each, HBM is 288 GB)
