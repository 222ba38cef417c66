#!/usr/bin/env python3
"""A/B of the page-ordered render kernel's experiment bits (PW_OPT_EXPERIMENT) on the C3 batch, all inside ONE
process on the same buffers (box-to-box and allocation-to-allocation spread exceeds most effects):

    bits 0..2  rotation of the page index inside groups of 8 consecutive pages (which XCD writes which page)
    bits 3..4  page order: 0 address order, 1 one contiguous eighth of the buffer per XCD, 2 every XCD writes runs
               of 2^(bits 24..28) consecutive pages (the chip-wide front stays 8 runs wide), 3 as 1 with XCD k starting
               k * (bits 32..47) pages into its eighth
    bits 20..22 cache policy of the observation stores (compile-time variants of the production kernel):
               0 nt (production), 1 plain, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt, 6 sc0 nt, 7 sc0
    bits 8..15 KiB of dynamic LDS per workgroup (occupancy limit)
    bit  7     no-op (selects the experiment build of the kernel with default behaviour)
    bit  16    per-environment page records (one scalar load on the fast path)

Every variant's output is compared byte for byte with the default's.  Usage: page_xp.py [--allocs N] [--reps R]"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--allocs", type=int, default=3)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--obs", default="uint8")
    args = ap.parse_args()
    B = 65536
    paths = bench.level1_paths()
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                       border_width=1, observation=args.obs, autoreset=True)
    vec.reset()
    g = torch.Generator(device=vec.device).manual_seed(1)
    for _ in range(30):  # a typical mid-episode batch
        vec.step(torch.randint(0, 4, (B,), generator=g, device=vec.device, dtype=torch.uint8))
    eng = vec.engine
    stride = eng.obs_stride

    def render_into(storage):
        _capi.check(_capi.lib.pw_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos), _capi._ptr(storage),
                                        stride, B, eng._stream()))

    def timed(storage, xp, reps):
        eng.set_option("experiment", xp)
        for _ in range(3):
            render_into(storage)
        eng.profile_render(reps)
        for _ in range(reps):
            render_into(storage)
        ms = np.array(eng.profile_read())
        eng.profile_render(0)
        return float(np.median(ms)), float(ms.min())

    esz = 1 if args.obs == "uint8" else 4
    ref = vec._obs_storage
    eng.set_option("experiment", 0)
    render_into(ref)
    torch.cuda.synchronize()
    variants = [("default", 0), ("xp-kernel-noop", 0x80), ("default-again", 0)]
    variants += [(f"rot{r}", r) for r in range(1, 8)]
    variants = [("default", 0), ("xcd-chunks", 1 << 3), ("runs of 64", (2 << 3) | (6 << 24))]
    variants += [(f"stagger {st}", (3 << 3) | (st << 32)) for st in (1, 3, 9, 33, 129, 585, 1171, 4097, 14464)]
    variants += [(f"stagger {st}+lds4K", (3 << 3) | (st << 32) | (4 << 8)) for st in (9, 585, 14464)]
    variants += [(f"stagger {st}+lds6K", (3 << 3) | (st << 32) | (6 << 8)) for st in (585,)]
    variants += [(f"stagger {st}+rec", (3 << 3) | (st << 32) | (1 << 16)) for st in (585,)]
    variants += [(f"lds{k}K", k << 8) for k in (5, 6, 7)]
    variants += [("runs64+lds6K", (2 << 3) | (6 << 24) | (6 << 8)), ("runs128+lds6K", (2 << 3) | (7 << 24) | (6 << 8)),
                 ("default-again", 0)]
    names = ["nt", "plain", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0 nt", "sc0"]
    if args.obs == "uint8":
        variants += [("store:" + names[k], k << 20) for k in (4,)]
    bufs = [("engine", ref)]
    for i in range(args.allocs):
        bufs.append((f"fresh{i}", torch.zeros((B, stride // esz), dtype=ref.dtype, device=vec.device)))
    print("%-16s" % "variant" + "".join("%18s" % n for n, _ in bufs))
    for name, xp in variants:
        row = []
        for bname, buf in bufs:
            med, mn = timed(buf, xp, args.reps)
            if buf is not ref:
                assert torch.equal(buf, ref), (name, bname)
            row.append("%9.4f/%8.4f" % (med, mn))
        # parity of the variant on the engine's buffer against the default's bytes in a fresh buffer
        print("%-16s" % name + "".join(row), flush=True)
    eng.set_option("experiment", 0)
    # fill_ of the same buffers (pure write ceiling of each allocation)
    row = []
    for bname, buf in bufs:
        evs = []
        for _ in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            buf.zero_()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        t = np.array([a.elapsed_time(b) for a, b in evs])[2:]
        row.append("%9.4f/%8.4f" % (np.median(t), t.min()))
    print("%-16s" % "torch zero_" + "".join(row))


if __name__ == "__main__":
    main()
