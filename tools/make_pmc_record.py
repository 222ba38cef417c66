#!/usr/bin/env python3
"""HBM bytes per launch of the dominant (render) kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
in separate runs with --kernel-trace only), corrected as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE
counts the 128-byte requests of a wide coalesced read as 64 bytes on gfx950, so it is doubled.

    python tools/make_pmc_record.py fetch_results.db write_results.db KERNEL_SUBSTRING envs obs_bytes source > record.json
"""
import hashlib
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCE = os.path.join(ROOT, "pushworld_amd", "csrc", "pw_render_kernels.inc")


def kernel_source_sha():
    """sha256 of the render kernels' source: bench.py copies a record into its line only for the source it was
    measured on."""
    with open(KERNEL_SOURCE, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:  # noqa: BLE001
        return None


def mean_counter(db, kernel, counter):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select k.name, avg(p.counter_value), count(*) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
        "where p.counter_name = ? group by k.name", (counter,)).fetchall()
    best = None
    for name, val, n in rows:
        # the production symbol only (kTag = 0), not the tuner's trial launches (kTag = 1)
        if kernel in name and ", 1>" not in name and ", 1," not in name and (best is None or n > best[2]):
            best = (name, val, n)
    if best is None:
        raise SystemExit(f"{counter}: no kernel matching {kernel!r} in {db}")
    return best


def main():
    fetch_db, write_db, kernel, envs, obs_bytes, source = sys.argv[1:7]
    fname, fetch_kb, fn = mean_counter(fetch_db, kernel, "FETCH_SIZE")
    wname, write_kb, wn = mean_counter(write_db, kernel, "WRITE_SIZE")
    rec = {
        "kernel": kernel,
        "kernel_symbol": fname,
        "envs": int(envs),
        "obs_bytes": int(obs_bytes),
        "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
        "write_size_kb": write_kb,
        "fetch_size_kb_raw": fetch_kb,
        "dispatches": [fn, wn],
        "source": source,
        "kernel_source_sha16": kernel_source_sha(),
        "git_head": git_head(),
        "note": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, --kernel-trace only), mean over the "
                "dispatches of the kernel on the C3 workload; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                "tallies 128-B requests as 64 B)",
    }
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
