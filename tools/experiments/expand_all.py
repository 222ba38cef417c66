#!/usr/bin/env python3
"""Round-4 sweep: pw_expand4 (automatic kernel / grid choice) on EVERY benchmark puzzle of levels 1-4 -- 4 M-state frontiers
(the first 4 M states of a breadth-first search; smaller state spaces repeated) cycling through > 640 MB of buffers.
One line per puzzle + the distribution:  python tools/experiments/expand_all.py > gpurun_out/expand_all.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import config_suite as cs  # noqa: E402


def main():
    from pushworld_amd.config import BENCHMARK_PUZZLES_PATH
    from pushworld_amd.utils.filesystem import iter_files_with_extension

    rels = sorted(os.path.relpath(f, BENCHMARK_PUZZLES_PATH) for f in iter_files_with_extension(BENCHMARK_PUZZLES_PATH, ".pwp")
                  if os.sep + "level0" + os.sep not in f and not os.path.relpath(f, BENCHMARK_PUZZLES_PATH).startswith("level0"))
    if len(sys.argv) > 1 and sys.argv[1].isdigit():
        rels = rels[::max(1, len(rels) // int(sys.argv[1]))]
    elif len(sys.argv) > 1:  # only the puzzles whose path contains one of the arguments
        rels = [r for r in rels if any(a in r for a in sys.argv[1:])]
    rows = []
    t_start = time.time()
    for rel in rels:
        try:
            pz, st_host, exhausted, distinct = cs.c5_frontier(rel, 4_000_000)
        except Exception as exc:  # noqa: BLE001
            print(f"{rel:48s} frontier failed: {exc!r}", flush=True)
            continue
        F, N = st_host.shape
        eng = pz._engine()
        dev = eng.device
        per = 20 * N + 20
        nbuf = max(1, -(-cs.CACHE_BUST_BYTES // (F * per)))
        sets = [(torch.as_tensor(st_host).to(dev), torch.empty((F, 4, N), dtype=torch.int32, device=dev),
                 torch.empty((F, 4), dtype=torch.int32, device=dev), torch.empty((F, 4), dtype=torch.uint8, device=dev)) for _ in range(nbuf)]
        it = [0]

        def one():
            s = sets[it[0] % nbuf]
            it[0] += 1
            eng.expand4(0, s[0], s[1], s[2], s[3])

        reps = max(12, 3 * nbuf)
        cs.wall(one, nbuf, nbuf)
        ms = cs.launch_ms(eng, one, reps)
        frac = F * per / (ms.mean() * 1e-3) / 1e9 / 8000
        rows.append((frac, rel, N, distinct))
        print(f"{rel:48s} N={N:2d} distinct {distinct:8d}  {ms.mean():7.4f} ms  {F / ms.mean() * 1e3:10.3e} parents/s  = {frac:.3f} of peak", flush=True)
        del sets
        torch.cuda.empty_cache()
    fr = np.array([r[0] for r in rows])
    q = np.quantile(fr, [0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
    print(f"# {len(rows)} puzzles in {time.time() - t_start:.0f} s: fraction of the 8 TB/s peak min / p10 / p25 / median / p75 / p90 / max = "
          + " / ".join(f"{v:.3f}" for v in q) + f"; at least 0.60: {int((fr >= 0.6).sum())}, at least 0.75: {int((fr >= 0.75).sum())}")
    for frac, rel, N, distinct in sorted(rows)[:12]:
        print(f"#   slowest: {rel:44s} N={N:2d} {frac:.3f}")


if __name__ == "__main__":
    main()
