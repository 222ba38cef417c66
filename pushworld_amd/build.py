"""Builds ``pushworld_amd/lib/libpushworld_amd.so`` for gfx950 with hipcc (in-tree).

    python -m pushworld_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpushworld_amd.so")
SOURCES = ["pw_host.cpp", "pw_kernels.hip"]
HEADERS = ["pw_host.h", "pw_format.h", "pw_zone.h", "pw_device_guard.h", "pw_search.inc", "pw_generate.inc", "pw_step_kernels.inc", "pw_seg_kernels.inc", "pw_mailbox_kernels.inc", "pw_expand_kernels.inc", "pw_mailbox.inc", "pw_render_kernels.inc",
           "pw_engine.inc", os.path.join("..", "..", "include", "pushworld_amd.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [
        hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-Wall", "-Wno-unused-function", "-fno-fast-math",
    ]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{proc.stdout}\n{proc.stderr}")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
