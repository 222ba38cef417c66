"""Access to the vendored benchmark puzzles (data of the reference repository:
``benchmark/puzzles/level{1..4}/*.pwp`` extracted, ``level0.zip`` as shipped)."""
from __future__ import annotations

import os
import zipfile
from typing import Dict, List, Optional

from .config import BENCHMARK_PUZZLES_PATH, PUZZLE_EXTENSION
from .puzzle import PushWorldPuzzle

LEVEL0_FAMILIES = ("all", "base", "goals", "obstacles", "shapes", "size", "walls")


def level_paths(level: int) -> List[str]:
    """Sorted ``.pwp`` paths of level 1..4."""
    d = os.path.join(BENCHMARK_PUZZLES_PATH, f"level{level}")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.lower().endswith(PUZZLE_EXTENSION)]


def level0_texts(families=LEVEL0_FAMILIES, split: str = "train", limit: Optional[int] = None) -> Dict[str, str]:
    """``{zip member: text}`` of level-0 puzzles, sorted by (family, index)."""
    out = {}
    with zipfile.ZipFile(os.path.join(BENCHMARK_PUZZLES_PATH, "level0.zip")) as z:
        names = set(z.namelist())
        for fam in families:
            i = 0
            while limit is None or i < limit:
                m = f"level0/{fam}/{split}/level_0_{fam}_{split}_{i}.pwp"
                if m not in names:
                    break
                out[m] = z.read(m).decode()
                i += 1
    return out


def load_level0(families=LEVEL0_FAMILIES, split: str = "train", limit: Optional[int] = None) -> List[PushWorldPuzzle]:
    return [PushWorldPuzzle(text=t) for t in level0_texts(families, split, limit).values()]


def load_levels(levels=(1, 2, 3, 4)) -> List[PushWorldPuzzle]:
    return [PushWorldPuzzle(p) for lv in levels for p in level_paths(lv)]
