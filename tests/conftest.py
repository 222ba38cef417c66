import json
import os
import sys
import zipfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "pushworld_amd", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the HIP library is a build product (git-ignored): cross-compile it for gfx950 when missing or stale
    from pushworld_amd import build as pw_build

    pw_build.build(force=False)


class GoldenData:
    """Fixtures generated from the reference by tests/golden/make_golden.py."""

    def __init__(self):
        with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
            self.meta = json.load(f)
        self.ref_render_hashes = self.meta.pop("_reference_test_rendering_hashes")
        self.traj = np.load(os.path.join(GOLDEN, "golden_traj.npz"))
        self.states = np.load(os.path.join(GOLDEN, "golden_states.npz"))
        self.images = np.load(os.path.join(GOLDEN, "golden_images.npz"))
        self._zip = None
        self.keys = sorted(self.meta)

    def text(self, key: str) -> str:
        kind, rel = key.split(":", 1)
        if kind == "rand":
            return self.meta[key]["text"]
        if kind == "bench":
            with open(os.path.join(DATA, "puzzles", rel)) as f:
                return f.read()
        if kind == "pytest":
            with open(os.path.join(ROOT, "tests", "puzzles", "ref_python", rel)) as f:
                return f.read()
        if kind == "cpptest":
            with open(os.path.join(ROOT, "tests", "puzzles", "ref_cpp", rel)) as f:
                return f.read()
        if kind == "l0":
            if self._zip is None:
                self._zip = zipfile.ZipFile(os.path.join(DATA, "puzzles", "level0.zip"))
            return self._zip.read(rel).decode()
        raise KeyError(key)

    def sequences(self, key: str):
        """Yields (name, actions, start_state or None, pos, reward, terminated, goals)."""
        for name in ("plan", "mid", "rand"):
            k = f"{key}|{name}|actions"
            if k in self.traj:
                start = self.traj[f"{key}|{name}|start"] if f"{key}|{name}|start" in self.traj else None
                yield (name, self.traj[k], start, self.traj[f"{key}|{name}|pos"], self.traj[f"{key}|{name}|reward"],
                       self.traj[f"{key}|{name}|terminated"], self.traj[f"{key}|{name}|goals"])


@pytest.fixture(scope="session")
def golden():
    return GoldenData()


def solution_plan(level: str, name: str):
    path = os.path.join(DATA, "solutions", level, name + ".yaml")
    with open(path) as f:
        for line in f:
            if line.startswith("plan:"):
                return ["LRUD".index(c) for c in line.split(":", 1)[1].strip()]
    raise ValueError(path)
