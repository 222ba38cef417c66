#!/usr/bin/env python3
"""HBM bytes per launch of the non-render kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs with
--kernel-trace only), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE counts the 128-byte requests of a wide coalesced
read as 64 bytes on gfx950: doubled), one record per configuration of tools/config_suite.py:

    python tools/make_kernel_pmc_record.py SOURCE  KEY:KERNEL_SUBSTRING:UNITS_PER_LAUNCH:FETCH_DB:WRITE_DB [...] > profiles/pmc_kernels_latest.json

(``KERNEL_SUBSTRING`` = ``a+b``: a call that launches two kernels -- the bound C4 rollouts: segments + listed lanes -- is their sum.)

``--merge=FILE`` first: records of FILE whose kernel sources (tools/config_suite.py: KEY_SOURCES) are unchanged are kept, so a
collection only has to re-measure what changed.  ``UNITS_PER_LAUNCH`` selects the dispatches (grid sizes differ between the single-step launches and the rollouts of one
profiled run: only dispatches whose duration-ordered position matches are not needed -- the caller passes one database pair per
launch shape).  The record carries the sha of pushworld_amd/csrc: bench.py copies a record into its line only for the source
it was measured on."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mean_counter(db, kernel, counter, grid=None):
    con = sqlite3.connect(db)
    q = ("select k.name, avg(p.counter_value), count(*), k.grid_x from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
         "where p.counter_name = ? group by k.name, k.grid_x")
    rows = [r for r in con.execute(q, (counter,)).fetchall() if kernel in r[0] and (grid is None or int(r[3]) == int(grid))]
    if not rows:
        return None
    total = sum(r[2] for r in rows)
    return max(rows, key=lambda r: r[2])[0], sum(r[1] * r[2] for r in rows) / total, total, sorted({int(r[3]) for r in rows})


def main():
    from tools.config_suite import csrc_sha, key_sha
    from tools.make_pmc_record import git_head

    source = sys.argv[1]
    configs = {}
    specs = sys.argv[2:]
    if specs and specs[0].startswith("--merge="):  # keep the records of an earlier file whose kernel sources are unchanged
        try:
            with open(specs[0].split("=", 1)[1]) as f:
                old = json.load(f)
        except (OSError, ValueError):
            old = {}
        for key, ent in old.get("configs", {}).items():
            if ent.get("source_sha16") == key_sha(key):
                configs[key] = ent
        specs = specs[1:]
    for spec in specs:
        parts = spec.split(":")
        key, kernel, units, fetch_db, write_db = parts[:5]
        grid = int(parts[5]) if len(parts) > 5 and parts[5] else None
        mult = int(parts[6]) if len(parts) > 6 and parts[6] else 1  # dispatches per call (a render in slices)
        if "+" in kernel:  # a call that launches several kernels (once each): the sum of their per-launch means
            fs = [mean_counter(fetch_db, k, "FETCH_SIZE", grid) for k in kernel.split("+")]
            ws = [mean_counter(write_db, k, "WRITE_SIZE", grid) for k in kernel.split("+")]
            f = None if any(x is None for x in fs) else (" + ".join(x[0] for x in fs), sum(x[1] for x in fs), min(x[2] for x in fs), sorted({g for x in fs for g in x[3]}))
            w = None if any(x is None for x in ws) else (" + ".join(x[0] for x in ws), sum(x[1] for x in ws), min(x[2] for x in ws), sorted({g for x in ws for g in x[3]}))
        else:
            f = mean_counter(fetch_db, kernel, "FETCH_SIZE", grid)
            w = mean_counter(write_db, kernel, "WRITE_SIZE", grid)
        if f is None or w is None:
            print(f"no dispatches of {kernel!r} (grid {grid}) in {fetch_db} / {write_db}", file=sys.stderr)
            continue
        configs[key] = {"kernel_symbol": f[0], "units_per_launch": int(units), "hbm_bytes_per_launch": (2.0 * f[1] + w[1]) * 1024.0 * mult,
                        "write_size_kb": w[1], "fetch_size_kb_raw": f[1], "dispatches": [f[2], w[2]], "grids": f[3],
                        "dispatches_per_call": mult, "source_sha16": key_sha(key), "collected_by": source, "git_head": git_head()}
    print(json.dumps({"configs": configs, "source": source, "csrc_sha16": csrc_sha(), "git_head": git_head(),
                      "note": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, --kernel-trace only), mean over the "
                              "dispatches of the kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B)"},
                     indent=1))


if __name__ == "__main__":
    main()
