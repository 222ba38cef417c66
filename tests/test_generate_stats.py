"""CPU: the level-0 generator's recipe pinned to the REFERENCE.  Bit equality is undefined (the reference
draws from Python's Mersenne Twister, this generator from a counter-based stream), so the pin is distributional:
tests/golden/golden_generator_stats.json holds histograms of 20 000 puzzles per configuration drawn with the imported
reference (tests/golden/make_generator_stats.py: width, height, wall / obstacle / goal-object counts, the shape index of
every role, the share of generate_puzzle attempts that failed); the host instance of the function the kernel runs
(pw_generate_level0, device = -1; tests/test_gpu_generate.py keeps device == host) must be statistically
indistinguishable: two-sample chi-square per histogram, p > 1e-3 (seeds are fixed, so the verdict is deterministic)."""
import json
import os

import numpy as np
import pytest
from scipy import stats

from pushworld_amd import generate

HERE = os.path.dirname(os.path.abspath(__file__))
SYM_W, SYM_A, SYM_M, SYM_G = 1, 3, 0x40, 0x80


@pytest.fixture(scope="module")
def ref_stats():
    with open(os.path.join(HERE, "golden", "golden_generator_stats.json")) as f:
        return json.load(f)


def chi2_p(a, b):
    """two-sample chi-square on a pair of histograms (bins empty in both are dropped)"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    keep = (a + b) > 0
    a, b = a[keep], b[keep]
    if len(a) < 2:
        return 1.0 if (a > 0).all() == (b > 0).all() else 0.0
    k1, k2 = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
    stat = (((k1 * a - k2 * b) ** 2) / (a + b)).sum()
    return float(stats.chi2.sf(stat, len(a) - 1))


def own_stats(kw, n, shapes, device=-1):
    """Histograms of ``n`` puzzles of this generator: the host instance (device -1) or the kernel (tests/test_gpu_generate.py)."""
    index = {frozenset(map(tuple, s)): i for i, s in enumerate(shapes)}
    grids, dims = generate.generate_level0_grids(n, random_seed=7, device=device, **kw)
    if device >= 0:
        grids, dims = grids.cpu().numpy(), dims.cpu().numpy()
    slot = kw.get("max_puzzle_size", 12)
    lo, hi = kw.get("min_puzzle_size", 8), slot
    n_shapes = 1 if kw.get("object_shapes") == "simple" else 9
    out = {k: np.zeros(m, np.int64) for k, m in (
        ("width", hi - lo + 1), ("height", hi - lo + 1),
        ("walls", kw.get("max_num_walls", 4) - kw.get("min_num_walls", 2) + 1),
        ("obstacles", kw.get("max_num_obstacles", 2) - kw.get("min_num_obstacles", 1) + 1),
        ("goals", kw.get("max_num_goal_objects", 1) - kw.get("min_num_goal_objects", 1) + 1),
        ("shape_m1", n_shapes), ("shape_m2", n_shapes), ("shape_agent", n_shapes), ("shape_obstacles", n_shapes))}
    for g, (w, h) in zip(grids, dims):
        g = g.reshape(slot, slot)
        ys, xs = np.nonzero(g)
        cells = {}
        for y, x in zip(ys.tolist(), xs.tolist()):
            cells.setdefault(int(g[y, x]), []).append((y, x))

        def shape_of(sym):
            c = cells[sym]
            y0, x0 = min(y for y, _ in c), min(x for _, x in c)
            return index[frozenset((y - y0, x - x0) for y, x in c)]  # (row, column) offsets like generate.py:215-225

        movers = sorted(s & 0x3f for s in cells if s & 0xc0 == SYM_M)
        n_goals = sum(1 for s in cells if s & 0xc0 == SYM_G)
        out["width"][w - lo] += 1
        out["height"][h - lo] += 1
        out["walls"][len(cells.get(SYM_W, ())) - kw.get("min_num_walls", 2)] += 1
        out["obstacles"][len(movers) - n_goals - kw.get("min_num_obstacles", 1)] += 1
        out["goals"][n_goals - kw.get("min_num_goal_objects", 1)] += 1
        out["shape_m1"][shape_of(SYM_M | 1)] += 1
        if n_goals == 2:
            out["shape_m2"][shape_of(SYM_M | 2)] += 1
        out["shape_agent"][shape_of(SYM_A)] += 1
        for k in movers:
            if k > n_goals:
                out["shape_obstacles"][shape_of(SYM_M | k)] += 1
    if device >= 0:  # (the attempt counter is a host-only entry point)
        return out, 0
    failed = generate.generate_level0_failed_attempts(n, random_seed=7, **kw)
    return out, int(failed.sum())


@pytest.mark.parametrize("name", ["default", "dense", "simple"])
def test_generator_distribution_matches_the_reference(ref_stats, name):
    ref = ref_stats[name]
    kw = ref["kwargs"]
    n = ref_stats["_n"]
    mine, failed = own_stats(kw, n, ref_stats["_shapes"])
    for key in ("width", "height", "walls", "obstacles", "goals", "shape_m1", "shape_m2", "shape_agent", "shape_obstacles"):
        assert mine[key].sum() > 0 or sum(ref[key]) == 0, key
        p = chi2_p(mine[key], ref[key])
        assert p > 1e-3, (name, key, p, mine[key].tolist(), ref[key])
    # share of generate_puzzle attempts that fail (the reference retries silently, generate.py:236-257): a two-proportion z test
    a1, f1 = n + failed, failed
    a2, f2 = ref["attempts"], ref["failed_attempts"]
    if f1 + f2 == 0:
        return
    pool = (f1 + f2) / (a1 + a2)
    z = (f1 / a1 - f2 / a2) / np.sqrt(pool * (1 - pool) * (1 / a1 + 1 / a2))
    assert abs(z) < 3.5, (name, f1, a1, f2, a2, z)


def test_the_statistics_tell_recipes_apart(ref_stats):
    """The test has teeth: a generator with another size range, or one that never rejects, is told apart."""
    ref = ref_stats["dense"]
    assert chi2_p(np.full(3, 20000 // 3), ref["width"]) < 1e-6                  # uniform sizes: no rejection bias
    other, _ = own_stats(dict(ref["kwargs"], max_num_obstacles=4), 4000, ref_stats["_shapes"])
    assert len(other["obstacles"]) == 5 and chi2_p(np.append(other["obstacles"], 0), ref["obstacles"]) < 1e-6
