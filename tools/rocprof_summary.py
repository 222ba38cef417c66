#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) as text.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [more.db ...] > profiles/r01_xxx.txt

Kernel table: calls, total / average / min / max duration, registers, LDS.  If the run
collected PMC counters, the per-kernel mean of every counter is listed as well.
"""
import sqlite3
import sys


def short(name, n=70):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def summarise(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f"# {path}")
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels "
        "group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} "
          f"{'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scr':>5s} {'grid':>10s} {'wg':>5s}")
    for r in rows:
        print(f"{short(r[0]):72s} {r[1]:6d} {r[2] / 1e3:11.1f} {r[3] / 1e3:10.2f} {r[4] / 1e3:10.2f} {r[5] / 1e3:10.2f} "
              f"{100.0 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:5d} {r[10]:10d} {r[11]:5d}")
    try:
        pm = cur.execute(
            "select k.name, p.counter_name, avg(p.counter_value), count(*) from pmc_events p join kernels k "
            "on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name order by k.name, p.counter_name"
        ).fetchall()
    except sqlite3.Error as exc:
        pm = []
        print("# (no counter data:", exc, ")")
    if pm:
        print()
        print(f"{'kernel':72s} {'counter':28s} {'mean per dispatch':>20s} {'n':>5s}")
        for name, ctr, val, n in pm:
            if name.startswith("void at::") or "rocclr" in name:
                continue
            print(f"{short(name):72s} {ctr:28s} {val:20.1f} {n:5d}")
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
