#!/usr/bin/env python3
"""Copies what tools/collect_profiles.sh left under gpurun_out/prof_<tag>/ (scratch) into profiles/ (tracked):

    python tools/publish_profiles.py r05

  <name>_summary.txt        -> profiles/<tag>_<name>.txt          (rocprofv3 kernel-trace / PMC summaries)
  bench.json, bench_full.json, trace_bench_line.json, bench_search.json
                            -> profiles/<tag>_bench.json (the <= 4 KB line), <tag>_bench_full.json, <tag>_trace_bench_line.json, <tag>_search.json
  pmc_kernels_latest.json, pmc_render_latest.json (when re-collected)   -> profiles/ (same names)
  bind_<config>_<variant>_{trace,fetch,write}_summary.txt (tools/collect_bind_profiles.sh) -> profiles/<tag>_bind_traffic.txt (one file)
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    n = 0
    for name in sorted(os.listdir(src)):
        p = os.path.join(src, name)
        if name.startswith("bind_"):
            continue  # (tools/collect_bind_profiles.sh: one file for the twelve passes, below)
        if name.endswith("_summary.txt") and os.path.getsize(p) > 0:
            shutil.copyfile(p, os.path.join(dst, f"{tag}_{name[:-len('_summary.txt')]}.txt"))
            n += 1
    parts = []
    for c in ("c4", "c3"):
        for v in ("unbound", "bound"):
            rows = []
            for f in ("trace", "fetch", "write"):
                q = os.path.join(src, f"bind_{c}_{v}_{f}_summary.txt")
                if os.path.exists(q):
                    with open(q) as fh:
                        rows += [l.rstrip()[:230] for l in fh if "pw_step" in l or "pw_bind" in l]
            if rows:
                parts.append(f"\n== {c} {v}\n" + "\n".join(rows))
    if parts:
        with open(os.path.join(dst, f"{tag}_bind_traffic.txt"), "w") as f:
            f.write(f"# rocprofv3 passes of tools/bench_bind.py (tools/collect_bind_profiles.sh {tag}): state-only stepping of 65 536 environments, "
                    "unbound vs bound (pw_batch_bind)\n# trace columns: calls, total us, avg us, min, max, %, vgpr, sgpr(?), lds, scratch, grid, wg; "
                    "FETCH_SIZE / WRITE_SIZE in KB per launch (HBM-side bytes = 2 x FETCH + WRITE, guide's gfx950 correction)\n")
            f.write("\n".join(parts) + "\n")
        n += 1
    for name, out in (("bench.json", f"{tag}_bench.json"), ("bench_full.json", f"{tag}_bench_full.json"),
                      ("trace_bench_line.json", f"{tag}_trace_bench_line.json"), ("bench_search.json", f"{tag}_search.json"),
                      ("pmc_kernels_latest.json", "pmc_kernels_latest.json"), ("pmc_render_latest.json", "pmc_render_latest.json")):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 2:
            if name == "bench.json":  # only the JSON line (stdout may carry warnings of libraries)
                with open(p) as f:
                    lines = [l for l in f if l.startswith("{") and '"metric"' in l]
                if lines:
                    with open(os.path.join(dst, out), "w") as f:
                        f.write(lines[-1])
                    n += 1
                continue
            shutil.copyfile(p, os.path.join(dst, out))
            n += 1
    print(f"{n} files published to profiles/ as {tag}_*")


if __name__ == "__main__":
    main()
