"""-m gpu: BOUND batches (pw_batch_bind, csrc/pw_seg_kernels.inc) against the C oracle (puzzle.py:348-411, gym_env.py:201-226).
A bound call steps the puzzles that enough environments of the batch play one lane per environment with the puzzle's push tables
in LDS (segments of up to 256 environments of one puzzle); the other environments keep the lane groups, in the same launch.  Here:
every step of every environment -- digest of the position row, float64 reward bits, terminated, truncated, step counter -- for
batches sorted by puzzle (consecutive segments), shuffled ones (segments through the index list), a C4-like mix whose Level-0 half
stays unbound, single steps and 64-step launches, one launch and two; the segment bookkeeping pw_batch_bind reports; re-binding
behind pw_resample; the mismatch counter; the page records a bound pw_step_render hands to the render."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
SOLUTIONS = os.path.join(ROOT, "pushworld_amd", "data", "solutions")


def _plan(level, path):
    name = os.path.splitext(os.path.basename(path))[0]
    with open(os.path.join(SOLUTIONS, level, name + ".yaml")) as f:
        for line in f:
            if line.startswith("plan:"):
                return np.array(["LRUD".index(c) for c in line.split(":", 1)[1].strip()], np.uint8)
    raise ValueError(path)


def _pool(kind):
    """(texts, plans {pool index: uint8 plan}, per-puzzle weights of the assignment)"""
    from pushworld_amd import benchmark_data as bd

    texts, plans = [], {}
    if kind == "level1":
        levels, n_l0 = (1,), 0
    elif kind == "levels":  # all 223 Level 1-4 puzzles (N_pad 32; `Mind The Gap` has no block: it stays with the lane groups)
        levels, n_l0 = (1, 2, 3, 4), 0
    elif kind == "l0only":  # 700 Level-0 puzzles, a few environments each: no segment at all
        levels, n_l0 = (), 700
    else:  # "c4": 700 Level-0 puzzles (two or three environments each: unbound) + the 223 Level 1-4 puzzles
        levels, n_l0 = (1, 2, 3, 4), 700
    if n_l0:
        texts += list(bd.level0_texts(limit=n_l0 // 7).values())
    first = len(texts)
    for lv in levels:
        for p in bd.level_paths(lv):
            plans[len(texts)] = _plan(f"level{lv}", p)
            with open(p) as f:
                texts.append(f.read())
    return texts, plans, first


def _ids(kind, B, n, first, rng, order):
    if kind == "l0only":
        ids = rng.integers(0, n, size=B)
    elif kind == "c4":  # half of the environments on the Level-0 puzzles, half on the Level 1-4 ones
        ids = np.concatenate([rng.integers(0, first, size=B // 2), first + (np.arange(B - B // 2) * (n - first)) // (B - B // 2)])
    else:
        ids = (np.arange(B, dtype=np.int64) * n) // B
    ids = np.sort(ids)
    if order == "shuffled":
        ids = rng.permutation(ids)
    elif order == "runs":  # runs of 40 environments per puzzle, every puzzle in several runs: bound, but not consecutive
        ids = ids.reshape(-1, 40)[rng.permutation(len(ids) // 40)].reshape(-1)
    return ids


def _actions(rng, ids, plans, T, every):
    B = len(ids)
    acts = rng.integers(0, 4, size=(T, B), dtype=np.uint8)
    driven = 0
    for b in range(0, B, every):
        plan = plans.get(int(ids[b]))
        if plan is None:
            continue
        cyc = np.concatenate([plan, np.zeros(1, np.uint8)])
        acts[:, b] = cyc[np.arange(T) % len(cyc)]
        driven += 1
    return acts, driven


def _oracle_run(texts, ids, acts, max_steps, NP):
    from oracle import c_oracle

    used = np.unique(ids)
    remap = np.full(len(texts), -1, np.int64)
    remap[used] = np.arange(len(used))
    oracles = [c_oracle.COraclePuzzle(texts[int(p)]) for p in used]
    w = np.random.default_rng(5).integers(-2**62, 2**62, size=NP * 2, dtype=np.int64)
    return w, c_oracle.rollout_digest(oracles, remap[ids], acts, max_steps, True, NP, w)


@pytest.mark.parametrize("kind,order,B,opts", [
    ("level1", "sorted", 8192, {}),
    ("level1", "shuffled", 8192, {}),
    ("level1", "runs", 8160, {"bind_min_envs": 1}),
    ("levels", "sorted", 16384, {}),
    ("levels", "shuffled", 16384, {"bind_fused": 2}),
    ("c4", "sorted", 32768, {}),
    ("c4", "shuffled", 32768, {}),
    ("c4", "sorted", 32768, {"bind_fused": 2}),
    ("c4", "sorted", 20000, {"step_quad16": "never", "bind_min_envs": 32}),
    ("levels", "sorted", 16384, {"bind_lanes": 1}),   # one lane per environment whatever the puzzle
    ("levels", "shuffled", 16384, {"bind_lanes": 2}),  # at most two
    ("levels", "shuffled", 16384, {"bind_spread": 3}),  # at most 16 environments per wavefront
    ("levels", "sorted", 16384, {"bind_lanes": 3}),    # at most four lanes: `Clean Sweep` (19 movables) two blocks of pairs per lane, not eight lanes
    ("c4", "sorted", 32768, {"bind_spread": 5, "bind_lanes": 1}),  # ... 4, one lane each
])
def test_bound_steps_against_the_oracle(kind, order, B, opts):
    import torch

    from pushworld_amd import _capi
    from pushworld_amd.vec_env import VecPushWorld

    texts, plans, first = _pool(kind)
    rng = np.random.default_rng(B + len(order))
    ids = _ids(kind, B, len(texts), first, rng, order)
    T, max_steps = 96, 70
    acts, driven = _actions(rng, ids, plans, T, 3)
    assert driven >= B // 8
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    vec = VecPushWorld(pset, B, puzzle_ids=ids, max_steps=max_steps, observation=None, device=0, autoreset=True, bind=True,
                       engine_options=opts)
    NP = vec.num_objects_padded
    vec.reset()
    info = vec.bound_info
    # the bookkeeping: which puzzles are bound follows from the assignment alone
    cnt = np.bincount(ids, minlength=len(texts))
    assert vec.engine.get_option("bind_puzzles") >= len(texts) - 1  # (every puzzle but `Mind The Gap` has a block)
    min_envs = vec.engine.get_option("bind_min_envs")
    eligible = cnt >= min_envs
    assert info["bound_puzzles"] in (int(eligible.sum()), int(eligible.sum()) - 1), info  # (- 1: `Mind The Gap`, 37 KB of tables)
    assert info["bound_envs"] <= int(cnt[eligible].sum()) and info["bound_envs"] >= int(cnt[eligible].sum()) - int(cnt.max())
    assert info["segments"] >= info["bound_puzzles"]
    if order == "sorted":
        assert info["listed_puzzles"] == 0
    else:
        assert info["listed_puzzles"] > 0
    if kind == "c4":
        assert B // 3 < info["bound_envs"] <= B - B // 2  # the Level-0 half stays with the lane groups
    else:
        assert info["bound_envs"] >= B - int(cnt.max())

    w, (want_d, want_r, want_te, want_tr, want_steps, want_last) = _oracle_run(texts, ids, acts, max_steps, NP)
    w_dev = torch.as_tensor(w).to(vec.device)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    for t in range(T):
        _, r, te, tr = vec.step(acts_dev[t])
        d = (vec.pos.view(B, NP * 2).to(torch.int64) * w_dev).sum(dim=1).cpu().numpy()
        bad = np.nonzero(d != want_d[t])[0]
        assert bad.size == 0, (t, bad[:5], ids[bad[:5]])
        assert (r.cpu().numpy().view(np.uint64) == want_r[t].view(np.uint64)).all(), t
        assert (te.cpu().numpy() == want_te[t]).all() and (tr.cpu().numpy() == want_tr[t]).all(), t
        if t % 16 == 15:
            assert (vec.steps.cpu().numpy() == want_steps[t]).all(), t
    assert (vec.states() == want_last).all()
    assert int((want_r == 10.0).sum()) > 0 and int((want_r > 0.5).sum()) > int((want_r == 10.0).sum())
    c = vec.counters()
    assert c["env_steps"] == B * T and c["episodes_solved"] == int((want_te != 0).sum())
    assert c["episodes_ended"] == int(((want_te | want_tr) != 0).sum()) and c["bad_actions"] == 0
    assert vec.engine.get_option("bind_mismatches") == 0


@pytest.mark.parametrize("kind,order,B,opts", [
    ("level1", "sorted", 8192, {}),
    ("levels", "shuffled", 16384, {}),
    ("levels", "sorted", 16384, {"bind_lanes": 1}),
    ("c4", "sorted", 32768, {}),                      # segments + the rest one lane each, 64 puzzles per wavefront (pw_step_mseg_kernel)
    ("c4", "shuffled", 32768, {"bind_fused": 2}),     # ... one kernel after the other on the caller's stream
    ("c4", "sorted", 20000, {"bind_min_envs": 64}),   # partly bound AND a rest with big puzzles: the lane groups step every environment
    ("c4", "sorted", 20000, {"bind_min_envs": 64, "bind_rollouts": 1}),  # ... or segments and lane groups side by side on two streams
    ("l0only", "sorted", 16384, {}),                  # no segment: every environment through pw_step_mseg_kernel
    ("l0only", "sorted", 16384, {"bind_spread": 3}),  # ... 16 environments per wavefront
    ("c4", "shuffled", 32768, {"bind_spread": 4}),    # 8 environments per wavefront in both kernels
    ("levels", "sorted", 16384, {"bind_spread": 2}),
    ("levels", "shuffled", 16384, {"bind_lanes": 3}),
])
def test_bound_rollouts_against_the_oracle(kind, order, B, opts):
    """64-step launches (pw_rollout) of a bound batch with every step's history, twice in a row (the second launch starts from
    the state the first left in HBM)."""
    import torch

    from pushworld_amd import _capi
    from pushworld_amd.vec_env import VecPushWorld

    texts, plans, first = _pool(kind)
    rng = np.random.default_rng(B + 7)
    ids = _ids(kind, B, len(texts), first, rng, order)
    T, max_steps = 128, 50
    acts, _ = _actions(rng, ids, plans, T, 3)
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    vec = VecPushWorld(pset, B, puzzle_ids=ids, max_steps=max_steps, observation=None, device=0, autoreset=True, bind=True,
                       engine_options=opts)
    NP = vec.num_objects_padded
    vec.reset()
    info = vec.bound_info
    if kind == "l0only":
        assert info["bound_envs"] == 0 and info["lane_envs"] == B
    elif kind == "c4" and "bind_min_envs" not in opts:
        assert info["bound_envs"] > B // 3 and info["bound_envs"] + info["lane_envs"] == B  # every environment is a lane
    elif kind == "c4":
        assert 0 < info["bound_envs"] + info["lane_envs"] < B  # (Level 1-4 puzzles below 64 environments: too big for a lane's slot)
    else:
        assert info["bound_envs"] > B // 3
    w, (want_d, want_r, want_te, want_tr, want_steps, want_last) = _oracle_run(texts, ids, acts, max_steps, NP)
    w_dev = torch.as_tensor(w).to(vec.device)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    for half in range(2):
        sl = slice(64 * half, 64 * half + 64)
        rh, th, uh = vec.rollout(acts_dev[sl].contiguous(), history=True)
        assert (rh.cpu().numpy().view(np.uint64) == want_r[sl].view(np.uint64)).all()
        assert (th.cpu().numpy() == want_te[sl]).all() and (uh.cpu().numpy() == want_tr[sl]).all()
        d = (vec.pos.view(B, NP * 2).to(torch.int64) * w_dev).sum(dim=1).cpu().numpy()
        assert (d == want_d[64 * half + 63]).all()
        assert (vec.steps.cpu().numpy() == want_steps[64 * half + 63]).all()
        assert (vec.reward.cpu().numpy().view(np.uint64) == want_r[64 * half + 63].view(np.uint64)).all()
    assert (vec.states() == want_last).all()
    assert vec.counters()["env_steps"] == B * T and vec.engine.get_option("bind_mismatches") == 0


def test_bound_step_render_hands_the_page_records_to_the_render():
    """pw_step_render on a bound batch: the segment kernel writes the page records the page-ordered render reads -- observations
    after 40 steps against the oracle's painter (puzzle.py:426-469), every environment's state row against its dynamics."""
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts, plans, _ = _pool("level1")
    B, T, max_steps = 8192, 40, 30
    rng = np.random.default_rng(77)
    ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
    acts, _ = _actions(rng, ids, plans, T, 3)
    vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=max_steps, pixels_per_cell=3,
                       border_width=1, observation="uint8", device=0, autoreset=True, fused=True, tune=False, bind=True)
    vec.reset()
    assert vec.bound_info["bound_envs"] == B
    NP = vec.num_objects_padded
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    w = np.random.default_rng(5).integers(-2**62, 2**62, size=NP * 2, dtype=np.int64)
    want_d, want_r, want_te, want_tr, _, want_last = c_oracle.rollout_digest(oracles, ids, acts, max_steps, True, NP, w)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    sel = np.unique(np.linspace(0, B - 1, 512).astype(np.int64))
    for t in range(T):
        obs, r, te, tr = vec.step(acts_dev[t])
        assert (r.cpu().numpy().view(np.uint64) == want_r[t].view(np.uint64)).all(), t
        assert (te.cpu().numpy() == want_te[t]).all() and (tr.cpu().numpy() == want_tr[t]).all(), t
        if t % 13 == 12 or t == T - 1:
            got = obs[torch.as_tensor(sel).to(vec.device)].cpu().numpy()
            want = c_oracle.observe_batch(oracles, ids, vec.states(), sel, 51, 42, 3, 1)
            assert (got == want).all(), t
    assert (vec.states() == want_last).all()


def test_rebinding_behind_resample_and_the_mismatch_counter():
    """Device-side episode turnover (pw_resample changes the ids of finished environments) on a bound batch: the binding is rebuilt
    behind every resample, on the stream; a twin without binding sees the same episodes.  Then the ids are changed BEHIND the
    binding's back: the launches stay memory-safe, play the puzzle the environments were bound to, and count them."""
    import torch

    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts, _, _ = _pool("level1")
    B, T = 6000, 150
    mk = lambda bind: VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, max_steps=20, observation=None, device=0,
                                   autoreset=True, resample=True, seed=9, bind=bind)
    a, b = mk(True), mk(False)
    a.reset()
    b.reset()
    assert a.bound_info["bound_envs"] > B // 2 and b.bound_info is None
    acts = torch.as_tensor(np.random.default_rng(1).integers(0, 4, size=(T, B), dtype=np.uint8)).to(a.device)
    for t in range(T):
        a.step(acts[t])
        b.step(acts[t])
        if t % 10 == 9 or t == T - 1:
            assert torch.equal(a.puzzle_id, b.puzzle_id) and torch.equal(a.pos, b.pos) and torch.equal(a.steps, b.steps), t
            assert torch.equal(a.reward.view(torch.int64), b.reward.view(torch.int64)), t
            assert torch.equal(a.terminated, b.terminated) and torch.equal(a.truncated, b.truncated), t
    assert a.counters() == b.counters() and a.counters()["episodes_ended"] > 5 * B
    assert a.engine.get_option("bind_mismatches") == 0
    # ... and a promise broken: ids written without telling the engine
    c = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, max_steps=20, observation=None, device=0, autoreset=True, bind=True)
    c.reset()
    c.puzzle_id.copy_(torch.roll(c.puzzle_id, 1))  # (not through set_puzzle_ids, which binds again)
    c.step(acts[0])
    torch.cuda.synchronize()
    n = c.engine.get_option("bind_mismatches")
    assert 0 < n <= B
    c.engine.bind(c.puzzle_id)  # binding again heals it
    c.reset()
    c.step(acts[1])
    assert c.engine.get_option("bind_mismatches") == n


def test_unbound_calls_keep_working_next_to_a_binding():
    """One binding per engine, keyed on the puzzle_id buffer and the batch size: calls on other buffers take the unbound launches,
    pw_batch_unbind ends it, and an engine without overlap tables binds nothing."""
    import torch

    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts, _, _ = _pool("level1")
    B = 4096
    ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
    pz = [PushWorldPuzzle(text=t) for t in texts]
    a = VecPushWorld(pz, B, puzzle_ids=ids, max_steps=25, observation=None, device=0, autoreset=True, bind=True)
    b = VecPushWorld(pz, B, puzzle_ids=ids, max_steps=25, observation=None, device=0, autoreset=True, bind=False)
    a.reset()
    b.reset()
    acts = torch.as_tensor(np.random.default_rng(2).integers(0, 4, size=(60, B), dtype=np.uint8)).to(a.device)
    other_ids = a.puzzle_id.clone()  # the same ids in ANOTHER buffer: an unbound call on the bound engine
    for t in range(60):
        if t % 3 == 0:
            a.engine.step(other_ids, acts[t], a.pos, a.steps, a.reward, a.dgoals, a.terminated, a.truncated, a.flags)
        elif t == 31:
            a.engine.unbind()
            a.step(acts[t])
        elif t == 41:
            assert a.engine.bind(a.puzzle_id)["bound_envs"] == B
            a.step(acts[t])
        else:
            a.step(acts[t])
        b.step(acts[t])
    assert torch.equal(a.pos, b.pos) and torch.equal(a.steps, b.steps) and torch.equal(a.terminated, b.terminated)
    assert torch.equal(a.reward.view(torch.int64), b.reward.view(torch.int64)) and a.counters() == b.counters()
    n = VecPushWorld(pz, B, puzzle_ids=ids, max_steps=25, observation=None, device=0, bind=True, engine_options={"step_tables": "none"})
    n.reset()
    assert n.bound_info == {"segments": 0, "bound_envs": 0, "bound_puzzles": 0, "listed_puzzles": 0, "lane_envs": 0}
    n.step(acts[0])


@pytest.mark.parametrize("full", [True, False])
def test_sets_of_8x8_puzzles_segments_when_every_environment_is_bound(full):
    """A set whose puzzles all fit 8 x 8 (the whole-grid board kernel's sets; C2 is one): a bound batch takes the segments when they hold
    EVERY environment and keeps the boards otherwise (a puzzle below the threshold) -- single steps and 64-step launches against the
    oracle either way."""
    import torch

    from pushworld_amd import _capi
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.vec_env import VecPushWorld

    texts = []
    for t in bd.level0_texts(limit=60).values():
        rows = [ln.split() for ln in t.strip().splitlines()]
        if len(rows) <= 6 and max(len(r) for r in rows) <= 6:  # (8 x 8 with the border)
            texts.append(t)
    texts = texts[:12]
    assert len(texts) >= 6
    B, T, max_steps = 4096, 96, 40
    rng = np.random.default_rng(3)
    ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
    if not full:  # the last puzzle keeps 25 environments: below the threshold of 48, not bound
        ids[np.nonzero(ids == len(texts) - 1)[0][25:]] = 1
    acts = rng.integers(0, 4, size=(T, B), dtype=np.uint8)
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    vec = VecPushWorld(pset, B, puzzle_ids=ids, max_steps=max_steps, observation=None, device=0, autoreset=True)  # (bind: the default)
    assert vec.engine.get_option("step_board_set") == 1
    NP = vec.num_objects_padded
    vec.reset()
    info = vec.bound_info
    assert info is not None and (info["bound_envs"] == B if full else 0 < info["bound_envs"] < B), info
    w, (want_d, want_r, want_te, want_tr, want_steps, want_last) = _oracle_run(texts, ids, acts, max_steps, NP)
    w_dev = torch.as_tensor(w).to(vec.device)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    for t in range(32):
        _, r, te, tr = vec.step(acts_dev[t])
        d = (vec.pos.view(B, NP * 2).to(torch.int64) * w_dev).sum(dim=1).cpu().numpy()
        assert (d == want_d[t]).all(), t
        assert (r.cpu().numpy().view(np.uint64) == want_r[t].view(np.uint64)).all(), t
        assert (te.cpu().numpy() == want_te[t]).all() and (tr.cpu().numpy() == want_tr[t]).all(), t
    r, te, tr = vec.rollout(acts_dev[32:].contiguous(), history=True)
    assert (r.cpu().numpy().view(np.uint64) == want_r[32:].view(np.uint64)).all()
    assert (te.cpu().numpy() == want_te[32:]).all() and (tr.cpu().numpy() == want_tr[32:]).all()
    assert (vec.states() == want_last).all()
    assert (vec.steps.cpu().numpy() == want_steps[T - 1]).all()
    c = vec.counters()
    assert c["env_steps"] == B * T and c["episodes_ended"] == int(((want_te | want_tr) != 0).sum())
