#!/usr/bin/env python3
"""Which PMC records of profiles/pmc_kernels_latest.json are stale (their kernel sources changed since they were measured):

    python tools/prof_state.py stale [KEY ...]      prints the stale ones among KEY ... (all keys without arguments), exit 0
    python tools/prof_state.py stamp KEY ...         one-time: marks the records of KEY ... as measured on the CURRENT sources
                                                     (for records collected before the per-record hashes existed, when `git log`
                                                     shows their sources untouched since)

tools/collect_profiles.sh asks before every group of rocprofv3 passes: a tweak to one kernel re-spends GPU minutes on that
kernel's configurations only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = os.path.join(ROOT, "profiles", "pmc_kernels_latest.json")


def main():
    from tools.config_suite import KEY_SOURCES, key_sha

    cmd, keys = sys.argv[1], sys.argv[2:] or list(KEY_SOURCES)
    try:
        with open(PATH) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        rec = {"configs": {}}
    if cmd == "stale":
        print(" ".join(k for k in keys if rec.get("configs", {}).get(k, {}).get("source_sha16") != key_sha(k)))
    elif cmd == "stamp":
        for k in keys:
            if k in rec.get("configs", {}):
                rec["configs"][k]["source_sha16"] = key_sha(k)
        with open(PATH, "w") as f:
            json.dump(rec, f, indent=1)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
