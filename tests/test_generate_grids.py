"""CPU: the level-0 generator behind pw_generate_level0 (host instance of the function the kernel runs) -- recipe
properties of generate.py:74-259, determinism, and the text form through the parser."""
import pytest

from pushworld_amd import _capi, generate

SYM_W, SYM_AW, SYM_A, SYM_M, SYM_G = 1, 2, 3, 0x40, 0x80


def _components(cells):
    """4-connected components of a set of (x, y) cells -- a tromino's cells are 8-connected at most"""
    cells, out = set(cells), []
    while cells:
        stack, comp = [cells.pop()], set()
        while stack:
            c = stack.pop()
            comp.add(c)
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    n = (c[0] + dx, c[1] + dy)
                    if n in cells:
                        cells.remove(n)
                        stack.append(n)
        out.append(comp)
    return out


@pytest.mark.parametrize("kw", [
    dict(),
    dict(min_puzzle_size=5, max_puzzle_size=7, min_num_walls=0, max_num_walls=9, min_num_obstacles=0, max_num_obstacles=5,
         min_num_goal_objects=1, max_num_goal_objects=2),
    dict(object_shapes="simple", min_puzzle_size=3, max_puzzle_size=4, max_num_walls=2),
])
def test_host_generator_follows_the_recipe(kw):
    n = 400
    grids, dims = generate.generate_level0_grids(n, random_seed=11, device=-1, **kw)
    again, dims2 = generate.generate_level0_grids(n, random_seed=11, device=-1, **kw)
    assert (grids == again).all() and (dims == dims2).all()                     # a pure function of (seed, index)
    tail, tdims = generate.generate_level0_grids(50, random_seed=11, device=-1, first=350, **kw)
    assert (tail == grids[350:]).all() and (tdims == dims[350:]).all()           # ... whatever the batch it is part of
    other, _ = generate.generate_level0_grids(n, random_seed=12, device=-1, **kw)
    assert (other != grids).any()
    lo, hi = kw.get("min_puzzle_size", 8), kw.get("max_puzzle_size", 12)
    slot = hi
    sizes, goals, walls, obst = set(), set(), set(), set()
    for g, (w, h) in zip(grids, dims):
        assert lo <= w <= hi and lo <= h <= hi
        g = g.reshape(slot, slot)
        assert not g[h:].any() and not g[:, w:].any()                            # nothing outside the puzzle's own grid
        by = {}
        for y in range(h):
            for x in range(w):
                if g[y, x]:
                    by.setdefault(int(g[y, x]), set()).add((x, y))
        assert SYM_A in by and 1 <= len(by[SYM_A]) <= 3
        n_goals = sum(1 for s in by if s & 0xc0 == SYM_G)
        assert 1 <= n_goals <= 2 and (SYM_M | 1) in by and (SYM_G | 1) in by
        for k in range(1, n_goals + 1):                                          # goal and its object: one shape
            m, t = by[SYM_M | k], by[SYM_G | k]
            mo, to = (min(x for x, _ in m), min(y for _, y in m)), (min(x for x, _ in t), min(y for _, y in t))
            assert {(x - mo[0], y - mo[1]) for x, y in m} == {(x - to[0], y - to[1]) for x, y in t}
        if n_goals == 2:                                                          # of different shapes (generate.py:108-115)
            a, b = by[SYM_M | 1], by[SYM_M | 2]
            norm = lambda s: {(x - min(p for p, _ in s), y - min(q for _, q in s)) for x, y in s}  # noqa: E731
            assert norm(a) != norm(b)
        movers = sorted(s & 0x3f for s in by if s & 0xc0 == SYM_M)
        assert movers == list(range(1, len(movers) + 1))                          # M1 .. Mk, obstacles numbered after the goals
        for s, cells in by.items():
            if s != SYM_W:
                assert len(cells) <= 3 and len(_components(cells)) == 1
            if kw.get("object_shapes") == "simple" and s != SYM_W:
                assert len(cells) == 1
        sizes.add((int(w), int(h)))
        goals.add(n_goals)
        walls.add(len(by.get(SYM_W, ())))
        obst.add(len(movers) - n_goals)
        # the text form parses to the same puzzle
        text = generate.grid_to_text(g.reshape(-1), w, h)
        rows = text.split("\n")
        assert len(rows) == h and all(len(r.split("  ")) == w for r in rows)
        p = _capi.ParsedPuzzle(text)
        assert (p.width, p.height, p.num_movables, p.num_goals) == (w + 2, h + 2, 1 + len(movers), n_goals)
    assert walls == set(range(kw.get("min_num_walls", 2), kw.get("max_num_walls", 4) + 1))
    assert obst == set(range(kw.get("min_num_obstacles", 1), kw.get("max_num_obstacles", 2) + 1))
    assert goals == set(range(kw.get("min_num_goal_objects", 1), kw.get("max_num_goal_objects", 1) + 1))
    assert len(sizes) > 3


def test_generator_argument_checks():
    for bad in (dict(min_puzzle_size=9, max_puzzle_size=8), dict(max_puzzle_size=63), dict(min_num_walls=-1),
                dict(min_num_obstacles=3, max_num_obstacles=2), dict(max_num_goal_objects=3), dict(min_num_goal_objects=0),
                dict(object_shapes="simple", max_num_goal_objects=2), dict(object_shapes="round")):
        with pytest.raises(ValueError):
            generate.generate_level0_grids(4, device=-1, **bad)
    # a range nothing fits into: every attempt fails, dims stay 0 0
    _, dims = generate.generate_level0_grids(3, device=-1, min_puzzle_size=1, max_puzzle_size=1, min_num_walls=3, max_num_walls=3)
    assert (dims == 0).all()
