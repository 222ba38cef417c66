#!/usr/bin/env python3
"""Parser fuzz vectors generated FROM THE REFERENCE (build container only): random puzzle texts, many of
them malformed, and what python3/src/pushworld/puzzle.py does with them -- the exception type, or the parse
products (dimensions, state, goals, object cells, walls).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_parser_fuzz_golden.py
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/python3/src")
import numpy as np  # noqa: E402
from pushworld.puzzle import PushWorldPuzzle  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))
TOKENS = [".", ".", ".", ".", ".", "W", "W", "AW", "A", "M1", "M2", "M3", "G1", "G2", "M1+G2", "A+G1", "M2+AW", "a", "m1",
          "g3", "w", "aw", "M10", "G10", "M1+G1", "A+AW", "W+AW", "M1+W", "M4", "G4"]
ODD = ["X", "M", "G", "1", "A1", "MM1", "M1+", "+", "M1++G1", "..", "M-1", "G1+G1", "M1.5", "Ma"]


def random_text(rng):
    rows, cols = int(rng.integers(1, 7)), int(rng.integers(1, 8))
    kind = rng.random()
    lines = []
    for r in range(rows):
        n = cols
        if kind < 0.12 and rng.random() < 0.4:
            n = max(0, cols + int(rng.integers(-2, 3)))  # ragged
        toks = []
        for _ in range(n):
            if kind > 0.88 and rng.random() < 0.08:
                toks.append(ODD[int(rng.integers(len(ODD)))])
            else:
                toks.append(TOKENS[int(rng.integers(len(TOKENS)))])
        sep = " " * int(rng.integers(1, 4)) if rng.random() < 0.8 else "\t"
        lines.append(sep.join(toks))
    if rng.random() < 0.15:
        lines.insert(int(rng.integers(0, len(lines) + 1)), "")  # blank line
    text = "\n".join(lines)
    if rng.random() < 0.7:
        text += "\n"
    if rng.random() < 0.05:
        text = text.replace("\n", "\r\n")
    return text


def outcome(text, tmp):
    path = os.path.join(tmp, "p.pwp")
    with open(path, "w", newline="") as f:
        f.write(text)
    try:
        p = PushWorldPuzzle(path)
    except Exception as exc:  # noqa: BLE001
        return {"error": type(exc).__name__}
    return {
        "dimensions": [int(v) for v in p.dimensions],
        "initial_state": [[int(v) for v in xy] for xy in p.initial_state],
        "goal_state": [[int(v) for v in xy] for xy in p.goal_state],
        "object_cells": [sorted([int(a), int(b)] for a, b in o.cells) for o in p.movable_objects],
        "walls": sorted([int(a), int(b)] for a, b in p.wall_positions),
        "agent_walls": sorted([int(a), int(b)] for a, b in p.agent_wall_positions),
    }


def main():
    rng = np.random.default_rng(424242)
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        while len(out) < 600:
            text = random_text(rng)
            out.append({"text": text, "result": outcome(text, tmp)})
    with open(os.path.join(HERE, "golden_parser_fuzz.json"), "w") as f:
        json.dump(out, f, indent=0)
    from collections import Counter
    print(Counter(o["result"].get("error", "ok") for o in out))


if __name__ == "__main__":
    main()
