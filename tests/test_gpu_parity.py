"""-m gpu: the HIP path (through the C ABI) against the golden fixtures captured from the
reference and against the oracle, on identical inputs.  Bit-exact everywhere (integer /
byte work; float32 observations are an exact 256-entry division table)."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def puzzles(golden):
    from pushworld_amd.puzzle import PushWorldPuzzle

    return {k: PushWorldPuzzle(text=golden.text(k)) for k in golden.keys}


def _groups(golden, puzzles=None):
    g = {"bench": [], "tests": [], "l0": [], "level1": [], "tiny": [], "bench+tests": []}
    for k in golden.keys:
        if not k.startswith("l0:"):  # benchmark puzzles (three sequences each) + test / random ones (two): with bind_min_envs 3 the
            g["bench+tests"].append(k)  # first are bound, the second go through the lane-group role of the same launch
        if puzzles is not None:  # grid with its border walls within 8 x 8, at most 8 movables: pw_step_board_kernel's sets
            w, h = puzzles[k].dimensions
            if w <= 8 and h <= 8 and puzzles[k].num_movables <= 8:  # (dimensions include the border)
                g["tiny"].append(k)
        if k.startswith("bench:level1/"):
            g["level1"].append(k)  # at most 11 movables: an N_pad 16 pool (also part of "bench")
        if k.startswith("bench:"):
            g["bench"].append(k)
        elif k.startswith("l0:"):
            g["l0"].append(k)
        else:
            g["tests"].append(k)
    return g


def _step_options(kernel):
    """engine options of a step-kernel flavour: "group" (default: row tables read from global memory for single
    steps; pools with more than 16 movables per puzzle: two per lane), "group-lds" (row tables staged in LDS, the
    default of multi-step rollouts), "group-wide" (32-lane groups), "lane", "wave"; "group-tables" / "group-notables":
    overlap tables (PW_OPT_STEP_TABLES) for every puzzle / for none (default: for puzzles with movables beyond 8 x 8)."""
    if kernel.startswith("bound"):
        # a BOUND batch (pw_batch_bind; VecPushWorld(bind=True)): segments of one puzzle each, one lane per environment, push tables
        # in LDS.  "bound": every puzzle of the batch bound (min_envs 1); "bound-split": segments and lane groups as two launches;
        # "bound-mixed": only the puzzles with at least 3 environments bound, the others through the lane-group role of the launch
        opts = {"step_boards": "never", "step_lds_tables": 2, "bind_min_envs": 3 if kernel == "bound-mixed" else 1,
                "bind_rollouts": 1}  # (launches of several steps: segments next to the lane groups also where only some are bound)
        if kernel == "bound-split":
            opts["bind_fused"] = 2
        return opts
    if kernel == "group-noquad":  # the defaults without the 16 x 16 whole-grid boards (the lane groups for every workgroup)
        return {"step_kernel": "group", "step_lds_tables": 2, "step_boards": "never", "step_quad16": "never"}
    if kernel == "group-narrow":  # N_pad 16 pools: 8 lanes per environment, two movables per lane
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 2, "step_narrow_groups": 1}
    if kernel == "group-unmixed":  # N_pad 8 / 16 pools: ONE choice of lanes for the whole set (default: per workgroup of 32 environments)
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 2, "step_mixed_groups": "never"}
    if kernel == "group-16lanes":  # N_pad 32 pools: 16-lane groups for every environment (default: 8-lane groups where N <= 16)
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 2, "step_narrow_groups": 2}
    if kernel == "group-tables":
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 2, "step_tables": "all"}
    if kernel == "group-bigtables":  # tables only for the puzzles with big movables: the kernel instance with both paths
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 2, "step_tables": "big"}
    if kernel == "group-bigtables-lds":
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 1, "step_tables": "big"}
    if kernel == "group-notables":
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 2, "step_tables": "none"}
    if kernel == "group-lds":  # (the table-only kernels have nothing to stage: row loops, rows in LDS)
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 1, "step_tables": "none"}
    if kernel == "group-wide":  # N_pad 32 pools: 32 lanes per environment instead of two movables per lane
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 2, "step_wide_groups": 1}
    if kernel == "boards":  # sets of 8 x 8 puzzles: whole-grid boards in registers (the default for state-only launches there)
        return {"step_kernel": "group", "step_lds_tables": 2}
    if kernel == "big-batch":  # what state-only launches of >= 131 072 environments pick by themselves (PW_OPT_STEP_LANE_BATCH)
        return {"step_boards": "never", "step_kernel": "group", "step_lds_tables": 2, "step_lane_batch": 1}
    if kernel == "lane-notables":  # one lane per environment with the row loops (the engine has no tables)
        return {"step_boards": "never", "step_kernel": "lane", "step_lds_tables": 2, "step_tables": "none"}
    # "group", "lane": the defaults (tables for every puzzle: the loop-free instances); "wave": rows only
    return {"step_kernel": kernel, "step_lds_tables": 2, "step_boards": "never"}


@pytest.mark.parametrize("group,kernel", [("bench", "group"), ("tests", "group"), ("l0", "group"),
                                          ("bench", "group-lds"), ("tests", "group-lds"), ("l0", "group-lds"),
                                          ("bench", "group-wide"), ("bench", "group-16lanes"),
                                          ("bench", "group-tables"), ("tests", "group-tables"), ("l0", "group-tables"),
                                          ("bench", "group-notables"), ("tests", "group-notables"),
                                          ("bench", "group-bigtables"), ("bench", "group-bigtables-lds"),
                                          ("level1", "group"), ("level1", "group-narrow"), ("level1", "group-lds"),
                                          ("level1", "group-unmixed"), ("l0", "group-unmixed"), ("tests", "group-unmixed"),
                                          ("bench", "lane"), ("tests", "lane"), ("l0", "lane"), ("level1", "lane"),
                                          ("bench", "lane-notables"), ("tests", "lane-notables"),
                                          ("bench", "big-batch"), ("l0", "big-batch"),
                                          ("tiny", "boards"), ("tiny", "group"),
                                          ("l0", "group-noquad"), ("tiny", "group-noquad"), ("level1", "group-noquad"), ("tests", "group-noquad"),
                                          ("bench", "wave"), ("tests", "wave"),
                                          ("bench", "bound"), ("tests", "bound"), ("l0", "bound"), ("level1", "bound"), ("tiny", "bound"),
                                          ("bench", "bound-split"), ("l0", "bound-split"), ("bench+tests", "bound-mixed")])
def test_trajectories_match_reference(golden, puzzles, torch_mod, group, kernel, monkeypatch):
    """Every golden sequence (human plan, mid-plan random walk, random walk) of every puzzle in
    one mixed batch: positions, float64 reward bits, terminated, truncated, step counter.
    pw_step has three kernels (16/32 lanes per env = default, one lane per env, one wavefront per
    env); all are checked."""
    torch = torch_mod
    from pushworld_amd.vec_env import VecPushWorld

    keys = _groups(golden, puzzles)[group]
    pool = [puzzles[k] for k in keys]
    assert len(keys) >= 20 or group != "tiny"
    envs = []  # (pool index, key, seq tuple)
    for pi, k in enumerate(keys):
        for seq in golden.sequences(k):
            envs.append((pi, k, seq))
    B = len(envs)
    T = max(len(e[2][1]) for e in envs)
    max_steps = 50
    vec = VecPushWorld(pool, B, puzzle_ids=[e[0] for e in envs], max_steps=max_steps, observation=None, device=0,
                       engine_options=_step_options(kernel), bind=kernel.startswith("bound"))
    if kernel.startswith("bound"):
        vec.reset()
        # (every puzzle has a block but `Mind The Gap`; "bound-mixed": the puzzles with all three sequences)
        if kernel == "bound-mixed":
            assert 0 < vec.bound_info["bound_envs"] < B, vec.bound_info
        else:
            assert vec.bound_info["bound_envs"] > B - 1 - 3 * 8, vec.bound_info
    elif group == "tiny":  # the set qualifies for the whole-grid boards; "boards" runs them, "group" the lane groups
        assert vec.engine.get_option("step_board_set") == 1
        assert vec.engine.get_option("step_boards") == (0 if kernel == "boards" else 2)
    vec.reset()
    NP = vec.num_objects_padded
    # start states
    pos0 = vec.states()
    for b, (pi, k, seq) in enumerate(envs):
        if seq[2] is not None:
            n = seq[2].shape[0]
            pos0[b, :n] = seq[2].astype(np.int8)
    vec.set_states(pos0)
    actions = np.zeros((T, B), np.uint8)
    lens = np.zeros((B,), np.int64)
    for b, (_, _, seq) in enumerate(envs):
        actions[: len(seq[1]), b] = seq[1]
        lens[b] = len(seq[1])
    acts_dev = torch.as_tensor(actions).to(vec.device)
    pos_hist = np.zeros((T, B, NP, 2), np.int8)
    rew_hist = np.zeros((T, B), np.float64)
    term_hist = np.zeros((T, B), np.uint8)
    trunc_hist = np.zeros((T, B), np.uint8)
    for t in range(T):
        _, r, te, tr = vec.step(acts_dev[t])
        pos_hist[t] = vec.pos.cpu().numpy()
        rew_hist[t] = r.cpu().numpy()
        term_hist[t] = te.cpu().numpy()
        trunc_hist[t] = tr.cpu().numpy()
    steps = vec.steps.cpu().numpy()
    assert (steps == T).all()
    for b, (pi, k, seq) in enumerate(envs):
        name, acts, start, pos, rew, term, goals = seq
        L, n = len(acts), pos.shape[1]
        assert (pos_hist[:L, b, :n] == pos.astype(np.int8)).all(), (k, name)
        assert (pos_hist[:L, b, n:] == 0).all()
        assert (rew_hist[:L, b].view(np.uint64) == rew.view(np.uint64)).all(), (k, name)
        assert (term_hist[:L, b] == term).all(), (k, name)
        want_trunc = (np.arange(1, L + 1) >= max_steps).astype(np.uint8)
        assert (trunc_hist[:L, b] == want_trunc).all(), (k, name)


@pytest.mark.parametrize("kernel", ["group", "group-noquad", "group-lds", "group-wide", "group-16lanes", "group-tables", "group-notables", "group-bigtables",
                                    "lane", "lane-notables", "wave", "boards", "bound", "bound-split"])
def test_random_overlapping_states(golden, puzzles, torch_mod, kernel, monkeypatch):
    """Random in-bounds states (objects may overlap each other and walls): all 4 successors
    equal the reference's table lookups (pins the not-already-overlapping clause)."""
    torch = torch_mod
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if f"{k}|in" in golden.states]
    if kernel == "boards":  # only sets of 8 x 8 puzzles run on whole-grid boards
        tiny = set(_groups(golden, puzzles)["tiny"])
        keys = [k for k in keys if k in tiny]
        assert len(keys) >= 10
    pool = [puzzles[k] for k in keys]
    ids, rows = [], []
    for pi, k in enumerate(keys):
        st = golden.states[f"{k}|in"]
        for s in range(st.shape[0]):
            ids.append(pi)
            rows.append((k, s))
    B = len(ids)
    vec = VecPushWorld(pool, B, puzzle_ids=ids, observation=None, device=0, engine_options=_step_options(kernel),
                       bind=kernel.startswith("bound"))
    vec.reset()
    NP = vec.num_objects_padded
    base = np.zeros((B, NP, 2), np.int8)
    for b, (k, s) in enumerate(rows):
        st = golden.states[f"{k}|in"][s]
        base[b, : st.shape[0]] = st.astype(np.int8)
    for a in range(4):
        vec.set_states(base)
        vec.step(torch.full((B,), a, dtype=torch.uint8, device=vec.device))
        got = vec.states()
        for b, (k, s) in enumerate(rows):
            want = golden.states[f"{k}|out"][s, a]
            assert (got[b, : want.shape[0]] == want.astype(np.int8)).all(), (k, s, a)


def _render_cases(golden, kinds):
    for k in golden.keys:
        if not k.startswith(kinds):
            continue
        for ent in golden.meta[k]["renders"]:
            yield k, ent


@pytest.mark.parametrize("ppc,bw", [(3, 1), (8, 2), (20, 2)])
def test_render_u8_digests(golden, puzzles, torch_mod, ppc, bw):
    """uint8 render of every golden (puzzle, state) equals the reference image (SHA-256)."""
    torch = torch_mod
    from pushworld_amd import _capi

    n = 0
    for k, ent in _render_cases(golden, ("bench:", "pytest:", "cpptest:", "l0:", "rand:")):
        if ent["ppc"] != ppc or ent["bw"] != bw:
            continue
        img = puzzles[k].render([tuple(p) for p in ent["state"]], border_width=bw, pixels_per_cell=ppc)
        m = golden.meta[k]
        assert img.shape == (m["height"] * ppc, m["width"] * ppc, 3) and img.dtype == np.uint8
        assert sha(img) == ent["u8"], (k, ent["seq"], ent["t"])
        n += 1
    assert n > 200


@pytest.mark.parametrize("pad", ["own", "l1", "std"])
@pytest.mark.parametrize("ppc,bw", [(3, 1), (8, 2), (20, 2)])
def test_observation_f32_digests(golden, puzzles, torch_mod, ppc, bw, pad):
    """Padded float32 observation (env_utils.render_observation_padded) equals the reference."""
    from pushworld_amd.utils.env_utils import render_observation_padded

    pads = {"own": None, "l1": (51, 42), "std": (54, 47)}
    n = 0
    for k, ent in _render_cases(golden, ("bench:", "pytest:", "rand:")):
        if ent["ppc"] != ppc or ent["bw"] != bw or f"f32_{pad}" not in ent:
            continue
        if ppc == 20 and pad != "own" and (n % 7):  # 12 MB frames: sample
            n += 1
            continue
        m = golden.meta[k]
        mh, mw = (m["height"], m["width"]) if pads[pad] is None else pads[pad]
        obs = render_observation_padded(puzzles[k], [tuple(p) for p in ent["state"]], mh, mw, ppc, bw)
        assert obs.dtype == np.float32 and obs.shape == (mh * ppc, mw * ppc, 3)
        assert sha(obs) == ent[f"f32_{pad}"], (k, ent["seq"], ent["t"], pad)
        n += 1
    assert n > 100


def test_full_small_images(golden, puzzles, torch_mod):
    for key in golden.images.files:
        parts = key.split("|")
        if parts[1] != "init":
            continue
        k, ppc, bw = parts[0], int(parts[2]), int(parts[3])
        img = puzzles[k].render(puzzles[k].initial_state, border_width=bw, pixels_per_cell=ppc)
        want = golden.images[key]
        assert img.shape == want.shape
        assert (img == want).all(), (key, np.argwhere(img != want)[:5])


def _assert_equals_oracle(vec, paths, ids, frame, ppc, bw, obs_kind, n=64):
    """``n`` complete observations of ``vec``'s current states, spread over the batch, against the C ORACLE (painter of
    puzzle.py:596-638 + padding of env_utils.py:65-91): pins whatever kernel / launch configuration ``vec`` renders with to the
    reference, not to another HIP kernel."""
    import torch

    from oracle import c_oracle

    if not hasattr(_assert_equals_oracle, "cache"):
        _assert_equals_oracle.cache = {}
    key = tuple(paths)
    if key not in _assert_equals_oracle.cache:
        _assert_equals_oracle.cache[key] = [c_oracle.COraclePuzzle(open(p).read()) for p in paths]
    oracles = _assert_equals_oracle.cache[key]
    B = vec.num_envs
    sel = np.unique(np.linspace(0, B - 1, min(n, B)).astype(np.int64))
    got = vec.obs[torch.as_tensor(sel).to(vec.device)].cpu().numpy()
    want = c_oracle.observe_batch(oracles, ids, vec.states(), sel, frame[0], frame[1], ppc, bw, "f32" if obs_kind == "float32" else "u8")
    assert got.dtype == want.dtype and got.shape == want.shape
    diff = np.nonzero((got != want).any(axis=(1, 2, 3)))[0]
    assert diff.size == 0, sel[diff[:5]]



def _overlapping_states_in_bounds(pos, paths, ids):
    """Every object at the position of its left neighbour in the state vector, clamped into the grid by its own bounding box:
    overlapping (illegal) states the painter order matters for, yet states the reference defines (puzzle.py:453-458 draws in
    bounds only)."""
    from oracle import c_oracle

    _assert_equals_oracle.cache = getattr(_assert_equals_oracle, "cache", {})
    key = tuple(paths)
    if key not in _assert_equals_oracle.cache:
        _assert_equals_oracle.cache[key] = [c_oracle.COraclePuzzle(open(p).read()) for p in paths]
    oracles = _assert_equals_oracle.cache[key]
    over = pos.copy()
    for b in range(pos.shape[0]):
        pz = oracles[int(ids[b])]
        for j in range(1, pz.num_movables):
            w = max(c[0] for c in pz.py.shapes[j]) + 1
            h = max(c[1] for c in pz.py.shapes[j]) + 1
            over[b, j, 0] = min(int(pos[b, j - 1, 0]), pz.width - w)
            over[b, j, 1] = min(int(pos[b, j - 1, 1]), pz.height - h)
    return over


# (page order, log2 run, KiB of LDS padding[, every page loads])
PAGE_CONFIGS = [None, (0, 0, 0), (1, 0, 0), (1, 0, 5), (1, 1, 3), (1, 2, 0), (1, 3, 1), (2, 0, 0), (2, 3, 7), (2, 6, 0),
                (2, 11, 2), (2, 20, 9), (0, 0, 0, 1), (1, 1, 7, 1), (2, 6, 7, 1)]


@pytest.mark.parametrize("obs_kind", ["uint8", "float32"])
@pytest.mark.parametrize("path", PAGE_CONFIGS)
def test_page_render_matches_lds_kernel(golden, torch_mod, path, obs_kind, monkeypatch):
    """The page-ordered render kernel (default for uint8 / ppc 3: static-image copy + LDS entry window)
    and the per-environment LDS kernel (engine option render_kernel = "lds") produce byte-identical observations
    on a mixed Level-1 batch along random walks and on overlapping (illegal) states."""
    torch = torch_mod
    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    B, T = 2048, 25
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B

    def make(opts):
        return VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                            border_width=1, observation=obs_kind, device=0, autoreset=True, engine_options=opts)

    # every launch configuration of the page kernel (page order, run length, occupancy padding; None = defaults)
    # and both producers of its page records: the step kernel (fused=True -> pw_step_render) and the pre-pass
    ref = make({"render_kernel": "lds"})
    alt = make({} if path is None else dict(zip(("page_order", "page_run_log2", "page_lds_pad_kb", "page_load_all"), path)))
    alt.fused = path is None or path[0] != 1
    assert ref.engine.render_kernel != "pw_render_page_kernel" and alt.engine.render_kernel == "pw_render_page_kernel"
    o_ref, o_alt = ref.reset(), alt.reset()
    assert torch.equal(o_ref, o_alt)
    gen = torch.Generator(device=ref.device)
    gen.manual_seed(5)
    acts = torch.randint(0, 4, (T, B), generator=gen, device=ref.device, dtype=torch.uint8)
    for t in range(T):
        o_ref = ref.step(acts[t])[0]
        o_alt = alt.step(acts[t])[0]
        assert torch.equal(ref.pos, alt.pos)
        assert torch.equal(o_ref, o_alt), t
    # ... and THIS launch configuration against the oracle itself: 64 complete observations (every configuration the tuner may
    # pick in production is pinned to the reference's painter, not only to another HIP kernel)
    _assert_equals_oracle(alt, paths, ids, (51, 42), 3, 1, obs_kind)
    # overlapping states: every object at the position of its left neighbour in the state vector (kept inside the grid)
    pos = _overlapping_states_in_bounds(ref.states(), paths, ids)
    for v in (ref, alt):
        v.set_states(pos)
    assert torch.equal(ref.render(), alt.render())
    _assert_equals_oracle(alt, paths, ids, (51, 42), 3, 1, obs_kind)
    # ... and pushed partly out of the grid (no reference semantics there: the two kernels must still agree)
    pos[:, 1:] = pos[:, :-1]
    for v in (ref, alt):
        v.set_states(pos)
    assert torch.equal(ref.render(), alt.render())
    # the tuner leaves correct observations behind and a configuration from its candidate list
    idx = alt.engine.tune_render(alt.puzzle_id, alt.pos, alt._obs_storage)
    assert 0 <= idx < 20 and torch.equal(alt._obs_storage, ref._obs_storage)
    assert torch.equal(ref.render(), alt.render())


@pytest.mark.parametrize("obs_kind,ppc,bw,pad,B,cfg", [
    ("float32", 20, 2, None, 40, None),          # the reference's default observation (10.3 MB each, 10 080-byte rows)
    ("float32", 20, 2, (54, 47), 24, (1, 1, 4)),  # standard padding (env_utils.py:25-41), quarters
    ("float32", 8, 2, None, 300, (0, 0, 0)),
    ("float32", 4, 1, (54, 47), 700, (2, 3, 7)),  # 2 256-byte rows: a page spans three pixel rows
    ("uint8", 8, 2, None, 700, None),             # 1 008-byte rows: five rows per page
    ("uint8", 16, 3, None, 200, (2, 6, 0)),
    ("uint8", 12, 5, (64, 64), 150, None),        # 2 304-byte rows, widest borders, the engine's largest frame
    # pixel rows that are NOT whole 16-byte chunks: unaligned source loads + one straddling chunk per row
    ("uint8", 20, 2, None, 100, None),            # 2 520-byte rows (the reference's rgb_array size, 2.6 MB each)
    ("uint8", 20, 2, (54, 47), 60, (1, 0, 7)),    # 2 820-byte rows
    ("uint8", 5, 1, None, 1500, (2, 2, 7)),       # 630-byte rows: seven rows per page
    ("uint8", 7, 3, (51, 43), 600, (0, 0, 0)),    # 903-byte rows (odd), border = whole cell but one pixel
    ("float32", 5, 2, (52, 43), 300, None),       # 2 580-byte rows (a multiple of 4, not of 16)
    ("float32", 3, 1, (54, 47), 60, None),        # ppc 3 float32 keeps its own page kernel (control case)
])
def test_rowpage_render_matches_lds_kernel(golden, torch_mod, obs_kind, ppc, bw, pad, B, cfg):
    """The row-page kernel (page-ordered render for frames whose pixel rows are whole 16-byte chunks: copies from
    per-puzzle static row tables, LDS window under the movables) against the per-environment LDS kernel: byte
    identical on a mixed Level-1 batch along random walks with autoreset, on overlapping (illegal) states, for both
    producers of the page records and several page orders."""
    torch = torch_mod
    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    T = 16
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B

    def make(opts, **kw):
        return VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=9, pixels_per_cell=ppc,
                            border_width=bw, observation=obs_kind, pad_cells=pad, device=0, autoreset=True,
                            engine_options=opts, tune=False, **kw)

    ref = make({"render_kernel": "lds"})
    alt = make({} if cfg is None else dict(zip(("page_order", "page_run_log2", "page_lds_pad_kb"), cfg)), fused=cfg is None)
    if ppc == 3:
        assert alt.engine.render_kernel == "pw_render_page_kernel"
    else:
        assert ref.engine.render_kernel == "pw_render_generic_kernel" and alt.engine.render_kernel == "pw_render_rowpage_kernel"
    o_ref, o_alt = ref.reset(), alt.reset()
    assert torch.equal(o_ref, o_alt)
    gen = torch.Generator(device=ref.device)
    gen.manual_seed(7)
    acts = torch.randint(0, 4, (T, B), generator=gen, device=ref.device, dtype=torch.uint8)
    for t in range(T):
        o_ref = ref.step(acts[t])[0]
        o_alt = alt.step(acts[t])[0]
        assert torch.equal(ref.pos, alt.pos)
        assert torch.equal(o_ref, o_alt), t
    frame = pad if pad is not None else (51, 42)
    _assert_equals_oracle(alt, paths, ids, frame, ppc, bw, obs_kind, n=24 if ppc >= 16 else 64)  # this configuration vs the oracle
    pos = _overlapping_states_in_bounds(ref.states(), paths, ids)  # overlapping states: every object at its left neighbour's position
    for v in (ref, alt):
        v.set_states(pos)
    assert torch.equal(ref.render(), alt.render())
    _assert_equals_oracle(alt, paths, ids, frame, ppc, bw, obs_kind, n=24 if ppc >= 16 else 64)
    pos[:, 1:] = pos[:, :-1]   # ... and pushed partly out of the grid (no reference semantics: the kernels must still agree)
    for v in (ref, alt):
        v.set_states(pos)
    assert torch.equal(ref.render(), alt.render())
    if B <= 64:  # the tuner on a small batch of big frames
        idx = alt.engine.tune_render(alt.puzzle_id, alt.pos, alt._obs_storage)
        assert 0 <= idx < 20 and torch.equal(alt._obs_storage, ref._obs_storage)


def test_page_render_in_slices(golden, torch_mod, monkeypatch):
    """Buffers beyond 2^31 chunks (32 GiB) are rendered in consecutive slices of whole environments; forced
    here with 333-environment slices on a 2 048-environment batch (slice bases are not page aligned)."""
    torch = torch_mod
    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    B = 2048
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    kw = dict(puzzle_ids=ids, max_steps=200, pixels_per_cell=3, border_width=1, observation="uint8", device=0, autoreset=True)
    whole = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, **kw)
    sliced = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, engine_options={"page_slice_envs": 333}, **kw)
    assert torch.equal(whole.reset(), sliced.reset())
    gen = torch.Generator(device=whole.device).manual_seed(12)
    for t in range(10):
        a = torch.randint(0, 4, (B,), generator=gen, device=whole.device, dtype=torch.uint8)
        assert torch.equal(whole.step(a)[0], sliced.step(a)[0]), t
        assert torch.equal(whole._obs_storage, sliced._obs_storage), t


@pytest.mark.parametrize("force_fused", ["1", "0"])
def test_fused_step_render_matches_reference(golden, puzzles, torch_mod, force_fused, monkeypatch):
    """pw_step_render on a mixed batch, both schedules (engine option fused_step_render = 1: ONE launch, step in
    wave 0 + per-environment render; 0: step kernel + page-ordered render): states, rewards,
    flags equal the golden trajectories and the observation equals the oracle's image of the
    reference state at every checked step (uint8, ppc 3, frame = batch maximum)."""
    torch = torch_mod
    from oracle import c_oracle
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("bench:level1/")][::3] + ["pytest:trivial_tool.pwp"]
    pool = [puzzles[k] for k in keys]
    envs = []
    for pi, k in enumerate(keys):
        for seq in golden.sequences(k):
            if seq[0] in ("plan", "rand"):
                envs.append((pi, k, seq))
    B = len(envs)
    T = 120
    vec = VecPushWorld(pool, B, puzzle_ids=[e[0] for e in envs], max_steps=None, pixels_per_cell=3, border_width=1,
                       observation="uint8", device=0, fused=True, engine_options={"fused_step_render": int(force_fused)})
    obs0 = vec.reset()
    oracles = {k: c_oracle.COraclePuzzle(golden.text(k)) for k in keys}
    fh, fw = vec.engine.obs_shape[0] // 3, vec.engine.obs_shape[1] // 3
    for b in (0, B // 2, B - 1):
        o = oracles[envs[b][1]]
        assert (obs0[b].cpu().numpy() == o.observation(o.initial_state, fh, fw, 3, 1, dtype="u8")).all()
    actions = np.zeros((T, B), np.uint8)
    for b, (_, _, seq) in enumerate(envs):
        n = min(T, len(seq[1]))
        actions[:n, b] = seq[1][:n]
    acts_dev = torch.as_tensor(actions).to(vec.device)
    for t in range(T):
        obs, r, te, tr = vec.step(acts_dev[t])
        pos = vec.pos.cpu().numpy()
        rr, tt = r.cpu().numpy(), te.cpu().numpy()
        img = obs.cpu().numpy() if t % 17 == 0 or t == T - 1 else None
        for b, (pi, k, seq) in enumerate(envs):
            if t >= len(seq[1]):
                continue
            want = seq[3][t]
            assert (pos[b, : want.shape[0]] == want.astype(np.int8)).all(), (k, seq[0], t)
            assert rr[b].view(np.uint64) == seq[4][t].view(np.uint64) and tt[b] == seq[5][t], (k, seq[0], t)
            if img is not None and b % 5 == 0:
                o = oracles[k]
                st = tuple(map(tuple, want.tolist()))
                assert (img[b] == o.observation(st, fh, fw, 3, 1, dtype="u8")).all(), (k, seq[0], t)


@pytest.mark.parametrize("kernel", ["group", "group-noquad", "group-lds", "group-wide", "group-16lanes", "group-notables", "group-bigtables", "group-level1",
                                    "group-narrow", "lane", "lane-notables", "big-batch", "boards", "bound", "bound-split", "bound-mixed"])
@pytest.mark.parametrize("autoreset", [False, True])
def test_rollout_equals_repeated_steps(golden, puzzles, torch_mod, autoreset, kernel, monkeypatch):
    """pw_rollout (T steps in one launch) == T pw_step launches: final state and every step's
    reward / terminated / truncated, on a mixed batch, with and without next-step autoreset; the
    per-step history of the plan sequences also equals the golden rewards of the reference.
    "group-level1" / "group-narrow": the Level-1 pool (N_pad 16), where launches of several steps run in 8-lane groups
    with two movables per lane (automatically with the table-only kernels; forced, also for the single steps)."""
    torch = torch_mod
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith(("bench:", "pytest:"))]
    if kernel in ("group-level1", "group-narrow"):
        keys = [k for k in golden.keys if k.startswith("bench:level1/")]
        kernel = "group" if kernel == "group-level1" else kernel
    if kernel == "boards":  # sets of 8 x 8 puzzles (whole-grid boards in registers; history and bad actions as everywhere)
        keys = _groups(golden, puzzles)["tiny"]
    pool = [puzzles[k] for k in keys]
    envs = []
    for pi, k in enumerate(keys):
        for seq in golden.sequences(k):
            if seq[0] in ("plan", "rand"):
                envs.append((pi, k, seq))
    B, T = len(envs), 150
    actions = np.zeros((T, B), np.uint8)
    for b, (_, _, seq) in enumerate(envs):
        n = min(T, len(seq[1]))
        actions[:n, b] = seq[1][:n]
        actions[n:, b] = (np.arange(T - n) + b) % 4
    acts = torch.as_tensor(actions).to("cuda:0")
    ids = [e[0] for e in envs]
    opts = _step_options(kernel)
    if kernel == "bound-mixed":
        opts["bind_min_envs"] = 2  # (benchmark puzzles: plan + random walk = two environments each -> bound; the others one)
    bind = kernel.startswith("bound")
    a = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=40, observation=None, device=0, autoreset=autoreset,
                     engine_options=opts, bind=bind)
    b_ = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=40, observation=None, device=0, autoreset=autoreset,
                      engine_options=opts, bind=bind)
    a.reset()
    b_.reset()
    rh, th, uh = a.rollout(acts, history=True)
    rr = np.zeros((T, B), np.float64)
    tt = np.zeros((T, B), np.uint8)
    uu = np.zeros((T, B), np.uint8)
    for t in range(T):
        _, r, te, tr = b_.step(acts[t])
        rr[t], tt[t], uu[t] = r.cpu().numpy(), te.cpu().numpy(), tr.cpu().numpy()
    assert (rh.cpu().numpy().view(np.uint64) == rr.view(np.uint64)).all()
    assert (th.cpu().numpy() == tt).all() and (uh.cpu().numpy() == uu).all()
    assert torch.equal(a.pos, b_.pos) and torch.equal(a.steps, b_.steps)
    assert torch.equal(a.reward, b_.reward) and torch.equal(a.terminated, b_.terminated)
    if not autoreset:
        for b, (_, k, seq) in enumerate(envs):
            n = min(T, len(seq[1]))
            assert (rr[:n, b].view(np.uint64) == seq[4][:n].view(np.uint64)).all(), k
            assert (tt[:n, b] == seq[5][:n]).all(), k
