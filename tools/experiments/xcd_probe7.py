#!/usr/bin/env python3
"""Seventh probe: buffers assembled from 2 MB physical chunks (HIP virtual-memory API, tools/experiments/vmm_alloc.hip)
against hipMalloc, C3 and C4 sizes, pure writes in the eighths order (TB/s).  Nothing is freed."""
import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

probe = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "bin", "libxcdprobe.so"))
probe.xcd_probe_parts.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
vmm = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "bin", "libvmm.so"))
vmm.vmm_alloc.restype = ctypes.c_int
vmm.vmm_alloc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64,
                          ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]


def tbps(ptr, nbytes):
    best = 0.0
    for pad in (0, 4096):
        ms = ctypes.c_float()
        assert probe.xcd_probe_parts(ctypes.c_void_p(ptr), nbytes // 4096, 8, 6, pad, ctypes.byref(ms)) == 0
        best = max(best, nbytes / (ms.value * 1e-3) / 1e12)
    return best


torch.zeros(1, device="cuda")
MB = 1 << 20
keep = []
for name, nbytes in (("C4 u8 4.50 GB", 65536 * 68608), ("C3 u8 3.79 GB", 65536 * 57856)):
    row = []
    for k in range(6):
        t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        keep.append(t)
        row.append(tbps(t.data_ptr(), nbytes))
    print("%-16s hipMalloc          : %s" % (name, " ".join("%.2f" % v for v in row)), flush=True)
    for chunk, label in ((2 * MB, "vmm 2 MB chunks   "), (32 * MB, "vmm 32 MB chunks  ")):
        row = []
        for k in range(6):
            p, h = ctypes.c_void_p(), ctypes.c_void_p()
            n = vmm.vmm_alloc(0, nbytes, chunk, 1, 0, ctypes.byref(p), ctypes.byref(h))
            assert n > 0, n
            row.append(tbps(p.value, nbytes))
        print("%-16s %s: %s" % (name, label, " ".join("%.2f" % v for v in row)), flush=True)
