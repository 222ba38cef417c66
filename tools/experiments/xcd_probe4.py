#!/usr/bin/env python3
"""Fourth probe: do the allocation classes exist for other buffer sizes?  Pure writes, eight contiguous parts (one per
XCD), ten fresh allocations per size; ms and TB/s."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "bin", "libxcdprobe.so"))
lib.xcd_probe_parts.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                ctypes.POINTER(ctypes.c_float)]


def parts(ptr, n_pages, P, pad, reps=6):
    ms = ctypes.c_float()
    assert lib.xcd_probe_parts(ctypes.c_void_p(ptr), n_pages, P, reps, pad, ctypes.byref(ms)) == 0
    return ms.value


sizes = {"C3 u8 (65536 x 57856 B)": 65536 * 57856, "C4 u8 (65536 x 68608 B)": 65536 * 68608, "1.0 GiB": 1 << 30,
         "2.5 GB": 2_500_000_000 // 4096 * 4096, "6.0 GB": 6_000_000_000 // 4096 * 4096, "15.2 GB (C3 f32)": 65536 * 231424}
for name, nbytes in sizes.items():
    keep = []
    row = []
    for k in range(10):
        t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        keep.append(t)
        ms = min(parts(t.data_ptr(), nbytes // 4096, 8, pad) for pad in (0, 4096))
        row.append(nbytes / (ms * 1e-3) / 1e12)
    print("%-28s TB/s per allocation: %s" % (name, " ".join("%.2f" % v for v in row)), flush=True)
    del keep, t
    torch.cuda.empty_cache()
