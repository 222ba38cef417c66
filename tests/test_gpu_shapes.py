"""-m gpu: the push closure on random SHAPES.  The lane-group kernels treat movables that fit into 8 x 8 cells
(one uint64 board, shifts and ANDs) differently from bigger ones (row loops); the shipped puzzles hold few objects
near that border.  Here: random puzzles whose objects have bounding boxes of 1 .. 10 cells either way (7, 8 and 9
over-represented), 3 .. 18 movables (8-, 16- and 32-lane groups, two movables per lane in the step kernel), random
in-bounds states in which objects may overlap each other and the walls -- all four successors, the moved-object
masks and the goal flags against the oracle's restatement of the reference's table lookups."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIDES = [1, 2, 3, 5, 7, 7, 8, 8, 8, 9, 9, 10]


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def random_puzzle(rng, n_movables, size=30):
    """A .pwp text: `size` x `size` cells, ~6 % walls, the agent + n_movables - 1 movables with random cell sets in
    random tight bounding boxes, one goal for M0."""
    grid = [["."] * size for _ in range(size)]
    used = np.zeros((size, size), bool)
    names = ["A"] + [f"M{k}" for k in range(n_movables - 1)]
    for name in names:
        for _ in range(200):
            w, h = int(rng.choice(SIDES)), int(rng.choice(SIDES))
            if n_movables > 10:  # many objects: keep them small enough to fit
                w, h = min(w, int(rng.integers(1, 5))), h
            cells = rng.random((h, w)) < 0.55
            cells[0, rng.integers(0, w)] = cells[h - 1, rng.integers(0, w)] = True   # tight box
            cells[rng.integers(0, h), 0] = cells[rng.integers(0, h), w - 1] = True
            x, y = int(rng.integers(0, size - w + 1)), int(rng.integers(0, size - h + 1))
            if not (used[y:y + h, x:x + w] & cells).any():
                used[y:y + h, x:x + w] |= cells
                for cy, cx in zip(*np.nonzero(cells)):
                    grid[y + cy][x + cx] = name
                break
        else:
            raise RuntimeError("could not place an object")
    free = [(y, x) for y in range(size) for x in range(size) if not used[y, x]]
    gy, gx = free[int(rng.integers(0, len(free)))]
    grid[gy][gx] = "G0"
    used[gy, gx] = True
    for y in range(size):
        for x in range(size):
            if not used[y, x] and rng.random() < 0.06:
                grid[y][x] = "W"
    return "\n".join("  ".join(row) for row in grid)


def random_states(rng, pz, count):
    """Position2D states with every object inside the grid (borders included), otherwise anywhere."""
    W, H = pz.dimensions  # includes the border walls: pw_validate_state's frame
    dims = []
    for obj in pz.movable_objects:
        xs = [c[0] for c in obj.cells]
        ys = [c[1] for c in obj.cells]
        dims.append((max(xs) + 1, max(ys) + 1))
    st = np.zeros((count, len(dims)), np.int32)
    for j, (w, h) in enumerate(dims):
        st[:, j] = rng.integers(0, W - w + 1, count) * 10000 + rng.integers(0, H - h + 1, count)
    return st


@pytest.mark.parametrize("n_movables,seed", [(3, 0), (5, 1), (8, 2), (9, 3), (12, 4), (16, 5), (18, 6), (18, 7)])
def test_random_shapes_expand4_matches_oracle(n_movables, seed):
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle

    rng = np.random.default_rng(1000 + seed)
    for rep in range(3):
        text = random_puzzle(rng, n_movables)
        pz = PushWorldPuzzle(text=text, order="cpp")
        oz = c_oracle.COraclePuzzle(text, order="cpp")
        assert pz.num_movables == n_movables
        # states near the initial one (pushes happen) and uniformly random ones (overlaps happen)
        init = np.array([x * 10000 + y for (x, y) in pz.initial_state], np.int32)
        near = np.repeat(init[None], 3000, 0)
        jitter = rng.integers(-2, 3, near.shape) * 10000 + rng.integers(-2, 3, near.shape)
        near = near + jitter * (rng.random(near.shape) < 0.3)
        far = random_states(rng, pz, 3000)
        W, H = pz.dimensions
        x, y = near // 10000, near % 10000
        for j, obj in enumerate(pz.movable_objects):
            w = max(c[0] for c in obj.cells) + 1
            h = max(c[1] for c in obj.cells) + 1
            x[:, j] = np.clip(x[:, j], 0, W - w)
            y[:, j] = np.clip(y[:, j], 0, H - h)
        states = np.concatenate([x * 10000 + y, far]).astype(np.int32)
        s, m, g = pz.expand4(states)
        ws, wm, wg = c_oracle.expand4_batch(oz, states)
        s, m, g = s.cpu().numpy(), m.cpu().numpy().astype(np.uint32), g.cpu().numpy()
        bad = np.nonzero((s != ws).any(axis=(1, 2)) | (m != wm).any(axis=1) | (g != wg).any(axis=1))[0]
        assert bad.size == 0, (n_movables, seed, rep, states[bad[0]].tolist(), s[bad[0]].tolist(), ws[bad[0]].tolist())
        assert (wm != 0).mean() > 0.05  # the sample does move things


@pytest.mark.parametrize("n_movables,options", [(5, {}), (12, {}), (18, {}), (18, {"step_wide_groups": 1}),
                                                (18, {"step_lds_tables": 1, "step_tables": "none"}), (5, {"step_tables": "none"}),
                                                (12, {"step_tables": "none"}), (18, {"step_tables": "none"}),
                                                (12, {"step_tables": "big"}), (12, {"step_narrow_groups": 1}),
                                                (12, {"step_narrow_groups": 1, "step_tables": "none"})])
def test_random_shapes_step_matches_oracle(torch_mod, n_movables, options):
    """The same states through pw_step (8 / 16 lanes, two movables per lane, 32 lanes, LDS-staged rows)."""
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    torch = torch_mod
    rng = np.random.default_rng(77 + n_movables)
    text = random_puzzle(rng, n_movables)
    pz = PushWorldPuzzle(text=text)               # Python object order
    oz = c_oracle.COraclePuzzle(text, order="python")
    B = 4096
    states = random_states(rng, pz, B)
    want, _, _ = c_oracle.expand4_batch(oz, states)
    vec = VecPushWorld([pz], B, observation=None, device=0, engine_options=options)
    vec.reset()
    NP = vec.num_objects_padded
    base = np.zeros((B, NP, 2), np.int8)
    base[:, :n_movables, 0] = states // 10000
    base[:, :n_movables, 1] = states % 10000
    for a in range(4):
        vec.set_states(base)
        vec.step(torch.full((B,), a, dtype=torch.uint8, device=vec.device))
        got = vec.pos.cpu().numpy().astype(np.int32)
        got = got[:, :n_movables, 0] * 10000 + got[:, :n_movables, 1]
        bad = np.nonzero((got != want[:, a]).any(axis=1))[0]
        assert bad.size == 0, (a, states[bad[0]].tolist(), got[bad[0]].tolist(), want[bad[0], a].tolist())


def test_maximum_sizes_64x64_grid_32_movables(torch_mod):
    """The engine's limits: a 62 x 62 file grid (64 x 64 with the border walls) with 32 movables --
    successors, moved masks and goal flags of random states, the step kernels and the uint8 / float32 observations
    (the 64-row frame is where the row bitboards and the page records run out of bits) against the oracle; one cell or
    one movable more is a ValueError."""
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    torch = torch_mod
    rng = np.random.default_rng(5)
    text = random_puzzle(rng, 32, size=62)
    pz = PushWorldPuzzle(text=text, order="cpp")
    assert pz.dimensions == (64, 64) and pz.num_movables == 32
    oz = c_oracle.COraclePuzzle(text, order="cpp")
    states = random_states(rng, pz, 4000)
    s, m, g = pz.expand4(states)
    ws, wm, wg = c_oracle.expand4_batch(oz, states)
    assert (s.cpu().numpy() == ws).all() and (m.cpu().numpy().astype(np.uint32) == wm).all() and (g.cpu().numpy() == wg).all()

    pzp = PushWorldPuzzle(text=text)
    ozp = c_oracle.COraclePuzzle(text, order="python")
    B = 512
    st = random_states(rng, pzp, B)
    want, _, _ = c_oracle.expand4_batch(ozp, st)
    base = np.zeros((B, 32, 2), np.int8)
    base[:, :, 0] = st // 10000
    base[:, :, 1] = st % 10000
    for options in ({}, {"step_wide_groups": 1}, {"step_kernel": "lane"}, {"step_kernel": "wave"}):
        vec = VecPushWorld([pzp], B, observation=None, device=0, engine_options=options)
        vec.reset()
        for a in range(4):
            vec.set_states(base)
            vec.step(torch.full((B,), a, dtype=torch.uint8, device=vec.device))
            got = vec.pos.cpu().numpy().astype(np.int32)
            assert (got[:, :, 0] * 10000 + got[:, :, 1] == want[:, a]).all(), (options, a)
    for obs, ppc, bw in (("uint8", 3, 1), ("float32", 3, 1), ("uint8", 5, 2)):
        vec = VecPushWorld([pzp], 64, observation=obs, pixels_per_cell=ppc, border_width=bw, device=0, tune=False)
        vec.reset()
        vec.set_states(base[:64])
        img = vec.render().cpu().numpy()
        wantimg = c_oracle.observe_batch([ozp], np.zeros(64, np.int32), base[:64], np.arange(64), 64, 64, ppc, bw,
                                         "f32" if obs == "float32" else "u8")
        assert (img == wantimg).all(), (obs, ppc, bw)

    for size, n in ((63, 5), (20, 33)):
        with pytest.raises(ValueError):
            PushWorldPuzzle(text=random_puzzle(rng, n, size=size))._puzzle_set
