#!/usr/bin/env python3
"""Sixth probe: pure writes against a COPY from a cache-resident source (what the render kernel does per page), eighths
order, on fresh 3.79 GB buffers (class = whatever the pure writes reach).  ms for the buffer."""
import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "bin", "libxcdprobe.so"))
lib.xcd_probe_parts.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
lib.xcd_probe_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.POINTER(ctypes.c_float)]


def write(ptr, n, pad):
    ms = ctypes.c_float()
    assert lib.xcd_probe_parts(ctypes.c_void_p(ptr), n, 8, 6, pad, ctypes.byref(ms)) == 0
    return ms.value


def copy(ptr, src, src_pages, n, mode, pad):
    ms = ctypes.c_float()
    assert lib.xcd_probe_copy(ctypes.c_void_p(ptr), ctypes.c_void_p(src), src_pages, n, mode, 6, pad, ctypes.byref(ms)) == 0
    return ms.value


nbytes = 65536 * 57856
n = nbytes // 4096
src = torch.randint(0, 255, (4 << 20,), dtype=torch.uint8, device="cuda")  # 512 KB of it are used: what ONE XCD reads of the 3.9 MB of static images in the eighths order
keep = []
print("%-12s %s" % ("buffer", "  ".join("%-22s" % h for h in ("write pad 0/4K/7K", "copy pad 0/4K/7K", "copy 2 pages pad 0/4K/7K/10K"))))
for k in range(8):
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    keep.append(t)
    w = [write(t.data_ptr(), n, p) for p in (0, 4096, 7168)]
    c1 = [copy(t.data_ptr(), src.data_ptr(), 128, n, 1, p) for p in (0, 4096, 7168)]
    c2 = [copy(t.data_ptr(), src.data_ptr(), 128, n, 2, p) for p in (0, 4096, 7168, 10240)]
    print("%-12s %s" % ("#%d" % k, "  ".join("%-22s" % " ".join("%.4f" % v for v in row) for row in (w, c1, c2))), flush=True)
