#!/usr/bin/env python3
"""Composes profiles/r04_expand4.txt -- the C5 kernel's record -- from the bench line, the rocprofv3 kernel traces and the PMC
record of the same code:  python tools/make_expand_record.py [bench.json ...] > profiles/r04_expand4.txt"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
KEYS = ("C5_2_obstacle", "C5_pull_dont_push", "C5_four_pistons")


def main():
    benches = sys.argv[1:] or [os.path.join(P, "r04_bench.json")]
    pm = json.load(open(os.path.join(P, "pmc_kernels_latest.json")))
    out = ["# pw_expand4 (C5) with the round's final kernel -- pw_expand4_v2_kernel<N, 1, true>: push tables in LDS, whole-line staging,",
           "# non-temporal stores, one tile per wavefront up to 8 movables / 8 persistent workgroups per CU and a store pipeline beyond, 24-bit arithmetic -- 4 M-state frontiers cycling",
           "# through > 640 MB of buffers (beyond the 256 MB Infinity Cache; `2 Obstacle` has 39 023 reachable states: tiled to 4 M rows, the",
           "# two others are 4 M DISTINCT states).  Algorithmic bytes per parent 20 N + 20.  peak = 8 TB/s.",
           "# (The kernel's first version: r04_expand4_first_version.txt; every variant and session: r04_expand4_variants.txt.)", "#"]
    for n, path in enumerate(benches):
        d = json.loads(open(path).read().strip().splitlines()[-1])
        out.append(f"# 1.{n + 1} bench line {os.path.relpath(path, ROOT)} -> configs (library HIP events around each launch; back to back = wall clock of the same loop)")
        for k in KEYS:
            v = d["configs"][k]
            alg = v["algorithmic_bytes_per_unit"] * v["units_per_launch"]
            tr = v.get("traffic")
            out.append(f"{k:20s} N={v['movables']:>2} {v['value']:.4e} parents/s  avg launch {v['avg_launch_ms']:.4f} ms ({v['launches_timed']} timed)  frac {v['frac']:.3f}"
                       f"  back to back {v['ms_per_launch']:.4f} ms = {alg / v['ms_per_launch'] / 1e6 / 8000:.3f}"
                       + (f"  traffic {tr / 1e6:.1f} MB = {tr / alg:.4f} x algorithmic" if tr else "  traffic: no record of this code yet")
                       + (f"  cpu_baseline {v['cpu_baseline']['value']:.3e} parents/s ({v['cpu_baseline']['cores']} threads)" if "cpu_baseline" in v else ""))
        out.append("#")
    out.append("# 2. rocprofv3 --kernel-trace --stats of tools/profile_kernels.py --what expand (r04_x2ob_trace.txt, r04_xpdp_trace.txt, r04_x4p_trace.txt)")
    for tag, n in (("x2ob", 3), ("xpdp", 6), ("x4p", 12)):
        for line in open(os.path.join(P, f"r04_{tag}_trace.txt")):
            if "pw_expand4_v2_kernel" in line:
                parts = line.split()
                avg, calls = float(parts[-10]), parts[-12]
                by = 4000000 * (20 * n + 20)
                out.append(f"N={n:2d} {calls} launches avg {avg:.2f} us -> {by / avg / 1e3:.1f} GB/s = {by / avg / 1e3 / 8000:.3f} of peak   | {line.strip()[:50]}")
    out += ["#", "# 3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; FETCH_SIZE doubled per MI355X_MICROARCH.md) -> pmc_kernels_latest.json"]
    for k in KEYS:
        v = pm["configs"][k]
        n = int(re.search(r"<(\d+),", v["kernel_symbol"]).group(1))
        by = 4000000 * (20 * n + 20)
        out.append(f"{k:20s} FETCH_SIZE {v['fetch_size_kb_raw']:.1f} KB (x2) + WRITE_SIZE {v['write_size_kb']:.1f} KB = {v['hbm_bytes_per_launch'] / 1e6:.2f} MB per launch"
                   f" = {v['hbm_bytes_per_launch'] / by:.4f} x algorithmic ({by / 1e6:.1f} MB)")
    out += ["#", "# 4. SQ counters: r04_x2ob_sq1.txt, r04_xpdp_sq1.txt, r04_x4p_sq1.txt, r04_x4p_sq2.txt"]
    print("\n".join(out))


if __name__ == "__main__":
    main()
