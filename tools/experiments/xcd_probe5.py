#!/usr/bin/env python3
"""Fifth probe: is it the DISTANCE between the eight write fronts?  Same buffers, eight contiguous parts, part size
(= distance between neighbouring fronts) varied by a few pages around n / 8.  TB/s over the pages written."""
import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "bin", "libxcdprobe.so"))
lib.xcd_probe_parts_per.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                    ctypes.c_int, ctypes.POINTER(ctypes.c_float)]


def run(ptr, n_pages, per, pad=4096, reps=6):
    ms = ctypes.c_float()
    assert lib.xcd_probe_parts_per(ctypes.c_void_p(ptr), n_pages, 8, per, reps, pad, ctypes.byref(ms)) == 0
    return min(n_pages, 8 * per) * 4096 / (ms.value * 1e-3) / 1e12


DELTAS = (0, -1, -2, -3, -4, -8, -16, -32, -64, -100, -128, -256, -512, -1000, -1024, -2048, -4096)
print("%-26s per(pages)  " % "buffer" + " ".join("%6d" % d for d in DELTAS))
for name, nbytes in (("C4 u8 4.50 GB", 65536 * 68608), ("1.0 GiB", 1 << 30), ("C3 u8 3.79 GB", 65536 * 57856), ("2.0 GiB", 2 << 30)):
    keep = []
    for k in range(4):
        t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        keep.append(t)
        n = nbytes // 4096
        per0 = (n + 7) // 8
        print("%-26s %9d   " % ("%s #%d" % (name, k), per0) + " ".join("%6.2f" % run(t.data_ptr(), n, per0 + d) for d in DELTAS), flush=True)
    del keep, t
    torch.cuda.empty_cache()
