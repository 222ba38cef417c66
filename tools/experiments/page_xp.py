#!/usr/bin/env python3
"""Launch configurations of the page-ordered render kernel on the C3 batch, all inside ONE process on several
buffers of identical size (box-to-box and allocation-to-allocation spread exceeds most effects):

    page_order       0 page = workgroup index, 1 one contiguous eighth of the buffer per XCD, 2 runs of
                     2^page_run_log2 pages per XCD
    page_lds_pad_kb  KiB of unused dynamic LDS per workgroup (occupancy cap = width of the write front)

Every configuration's output is compared byte for byte with the first one's, and the engine's tuner
(pw_engine_tune_render) is run on every buffer.  Usage: page_xp.py [--allocs N] [--reps R] [--obs uint8|float32]
Earlier rounds of this experiment (store cache policies, page rotation, staggered eighths, page records) are
summarised in profiles/r02_page_xp.txt."""
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
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--obs", default="uint8")
    args = ap.parse_args()
    B = 65536
    paths = bench.level1_paths()
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                       border_width=1, observation=args.obs, autoreset=True, tune=False)
    vec.reset()
    g = torch.Generator(device=vec.device).manual_seed(1)
    for _ in range(30):  # a typical mid-episode batch
        vec.step(torch.randint(0, 4, (B,), generator=g, device=vec.device, dtype=torch.uint8))
    eng = vec.engine
    stride = eng.obs_stride

    def render_into(storage):
        _capi.check(_capi.lib.pw_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos), _capi._ptr(storage),
                                        stride, B, eng._stream()))

    def timed(storage, cfg, reps):
        for k, v in zip(("page_order", "page_run_log2", "page_lds_pad_kb"), cfg):
            eng.set_option(k, v)
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
    default = tuple(eng.get_option(k) for k in ("page_order", "page_run_log2", "page_lds_pad_kb"))
    render_into(ref)
    torch.cuda.synchronize()
    variants = [("default %s" % (default,), default), ("identity", (0, 0, 0)), ("eighths", (1, 0, 0))]
    variants += [(f"runs64 pad{k}K", (2, 6, k)) for k in (0, 5, 6, 7, 8, 9)]
    variants += [(f"runs32 pad{k}K", (2, 5, k)) for k in (7,)]
    variants += [(f"runs128 pad{k}K", (2, 7, k)) for k in (7,)]
    variants += [(f"identity pad{k}K", (0, 0, k)) for k in (6, 7, 8)]
    variants += [(f"eighths pad{k}K", (1, 0, k)) for k in (4, 7)]
    variants += [(f"quarters pad{k}K", (1, 1, k)) for k in (5, 6, 7, 8, 9, 10)]
    variants += [(f"halves pad{k}K", (1, 2, k)) for k in (4, 7)]
    variants += [("default again", default)]
    bufs = [("engine", ref)]
    for i in range(args.allocs):
        bufs.append((f"fresh{i}", torch.zeros((B, stride // esz), dtype=ref.dtype, device=vec.device)))
    print("%-28s" % "configuration" + "".join("%18s" % n for n, _ in bufs))
    for name, cfg in variants:
        row = []
        for bname, buf in bufs:
            med, mn = timed(buf, cfg, args.reps)
            if buf is not ref:
                assert torch.equal(buf, ref), (name, bname)
            row.append("%9.4f/%8.4f" % (med, mn))
        print("%-28s" % name + "".join(row), flush=True)
    # the tuner's pick per buffer, and what it measures afterwards
    row = []
    for bname, buf in bufs:
        idx = _capi.check(_capi.lib.pw_engine_tune_render(eng.handle, _capi._ptr(vec.puzzle_id), _capi._ptr(vec.pos),
                                                          _capi._ptr(buf), stride, B, eng._stream()))
        cfg = tuple(eng.get_option(k) for k in ("page_order", "page_run_log2", "page_lds_pad_kb"))
        assert torch.equal(buf, ref), ("tuner", bname)
        med, mn = timed(buf, cfg, args.reps)
        row.append("%9.4f %8s" % (med, "#%d%s" % (idx, str(cfg).replace(" ", ""))))
    print("%-28s" % "tuner pick" + "".join(row))
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
    print("%-28s" % "torch zero_" + "".join(row))


if __name__ == "__main__":
    main()
