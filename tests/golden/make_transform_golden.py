#!/usr/bin/env python3
"""Golden vectors for pushworld_amd.transform, generated FROM THE REFERENCE (build container only):
the 8 dihedral transforms of a few puzzles (python3/src/pushworld/transform.py:21-48).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_transform_golden.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/python3/src")
from pushworld.transform import get_puzzle_transforms  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PUZZLES = [
    "tests/puzzles/ref_python/shortest_path_tool.pwp",
    "tests/puzzles/ref_python/trivial_obstacle.pwp",
    "pushworld_amd/data/puzzles/level1/2 Obstacle.pwp",
    "pushworld_amd/data/puzzles/level1/Choose Wisely.pwp",
    "pushworld_amd/data/puzzles/level2/Pull Dont Push.pwp",
    "pushworld_amd/data/puzzles/level4/Four Pistons.pwp",
]

out = {}
for rel in PUZZLES:
    with open(os.path.join(REPO, rel)) as f:
        out[rel] = get_puzzle_transforms(f.read())
with open(os.path.join(HERE, "golden_transforms.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print({k: list(v) for k, v in out.items()})
