"""-m gpu: DEEP parity at batch scale.  The full-size tests of test_gpu_configs.py run 9-10 steps; a uniform
random policy spends 99 % of its steps with the agent alone or blocked, so at 65 536 environments -- where the production kernel
selection applies (lanes per environment chosen per workgroup, the 16 x 16 boards, one lane per environment from 131 072 on) --
multi-object push chains, goal hits and reward 10.0 were barely exercised.  Here: C3 and one C4 shard for 208 steps with a third
(C3) / a quarter (C4) of the environments driven by the HUMAN SOLUTION PLANS of their puzzles (data/solutions: the plan, one
ignored action for the next-step autoreset, the plan again), the rest by uniform random actions; every step of every environment
against the C oracle (puzzle.py:348-411, gym_env.py:201-226): a 64-bit digest of the position row, the float64 reward bits,
terminated, truncated and the step counter; the complete rows after the last step.  Plus state-only runs of 131 072 (single
steps) and 262 144 environments (one 64-step launch) on the same footing."""
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


def _actions(rng, ids, plans, T, every):
    """uint8 [T, B]: environment b follows the solution plan of its puzzle when it has one and b % every == 0 (plan, one ignored
    action -- the step on which the solved episode is reset --, plan again), uniform random actions otherwise."""
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


def _run_against_digest(vec, oracles, ids, acts, max_steps, rollout=False):
    import torch

    from oracle import c_oracle

    T, B = acts.shape
    NP = vec.num_objects_padded
    w = np.random.default_rng(99).integers(-2**62, 2**62, size=NP * 2, dtype=np.int64)
    want_d, want_r, want_te, want_tr, want_steps, want_last = c_oracle.rollout_digest(oracles, ids, acts, max_steps, True, NP, w)
    w_dev = torch.as_tensor(w).to(vec.device)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    vec.reset()
    if rollout:
        rh, th, uh = vec.rollout(acts_dev, history=True)
        assert (rh.cpu().numpy().view(np.uint64) == want_r.view(np.uint64)).all()
        assert (th.cpu().numpy() == want_te).all() and (uh.cpu().numpy() == want_tr).all()
    else:
        for t in range(T):
            _, r, te, tr = vec.step(acts_dev[t])
            d = (vec.pos.view(B, NP * 2).to(torch.int64) * w_dev).sum(dim=1).cpu().numpy()
            bad = np.nonzero(d != want_d[t])[0]
            assert bad.size == 0, (t, bad[:5], vec.pos[int(bad[0])].cpu().numpy().tolist())
            assert (r.cpu().numpy().view(np.uint64) == want_r[t].view(np.uint64)).all(), t
            assert (te.cpu().numpy() == want_te[t]).all() and (tr.cpu().numpy() == want_tr[t]).all(), t
            if t % 16 == 15:
                assert (vec.steps.cpu().numpy() == want_steps[t]).all(), t
    assert (vec.states() == want_last).all()
    assert (vec.steps.cpu().numpy() == want_steps[-1]).all()
    return want_r, want_te, want_tr


def _c3_pool():
    import bench

    paths = bench.level1_paths()
    texts = [open(p).read() for p in paths]
    plans = {i: _plan("level1", p) for i, p in enumerate(paths)}
    return texts, plans


def test_c3_208_steps_with_solution_plans():
    """The headline workload's own path (pw_step_render: step kernel with page records + uint8 ppc-3 page render) for 208 steps."""
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts, plans = _c3_pool()
    B, T, max_steps = 65536, 208, 120
    ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
    acts, driven = _actions(np.random.default_rng(17), ids, plans, T, 3)
    assert driven >= B // 4
    vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=max_steps, pixels_per_cell=3,
                       border_width=1, observation="uint8", device=0, autoreset=True, fused=True, tune_allocations=1)
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    want_r, want_te, want_tr = _run_against_digest(vec, oracles, ids, acts, max_steps)
    solved = int((want_r == 10.0).sum())
    pushes = int((want_r > 0.5).sum())
    assert solved >= driven and pushes >= solved  # every plan-driven environment solved its puzzle at least once
    # the observations after 208 steps of dynamics (1 024 of them, complete)
    sel = np.unique(np.linspace(0, B - 1, 1024).astype(np.int64))
    got = vec.obs[torch.as_tensor(sel).to(vec.device)].cpu().numpy()
    want = c_oracle.observe_batch(oracles, ids, vec.states(), sel, 51, 42, 3, 1)
    assert (got == want).all()
    c = vec.counters()
    assert c["env_steps"] == B * T and c["episodes_solved"] == int((want_te != 0).sum())
    assert c["episodes_ended"] == int(((want_te | want_tr) != 0).sum())


@pytest.mark.parametrize("rank", [3])
def test_c4_shard_208_steps_with_solution_plans(rank):
    """One rank's shard of the 524 288-environment full mix (14 000 Level-0 + 223 Level 1-4 puzzles, N_pad 32), state only: the
    production selection -- 16 x 16 boards for the Level-0 workgroups, lane groups of 8 / 16 lanes for the others."""
    from oracle import c_oracle
    from pushworld_amd import _capi
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.sharding import c4_global_puzzle_ids, shard_puzzle_ids
    from pushworld_amd.vec_env import VecPushWorld

    texts = list(bd.level0_texts().values())
    n_l0 = len(texts)
    plans = {}
    for lv in (1, 2, 3, 4):
        for p in bd.level_paths(lv):
            plans[len(texts)] = _plan(f"level{lv}", p)
            with open(p) as f:
                texts.append(f.read())
    B, T, max_steps = 65536, 208, 150
    ids = np.sort(shard_puzzle_ids(c4_global_puzzle_ids(8 * B, n_l0, len(texts) - n_l0, 100), rank, 8))
    acts, driven = _actions(np.random.default_rng(100 + rank), ids, plans, T, 2)
    assert driven >= B // 5  # half of the Level 1-4 half
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    vec = VecPushWorld(pset, B, puzzle_ids=ids, max_steps=max_steps, observation=None, device=0, autoreset=True)
    assert vec.num_objects_padded == 32 and vec.engine.get_option("step_quad16_puzzles") >= 14000
    used = np.unique(ids)
    remap = np.full(len(texts), -1, np.int64)
    remap[used] = np.arange(len(used))
    oracles = [c_oracle.COraclePuzzle(texts[int(p)]) for p in used]
    want_r, want_te, want_tr = _run_against_digest(vec, oracles, remap[ids], acts, max_steps)
    assert int((want_r == 10.0).sum()) >= driven // 2  # (plans longer than max_steps - 1 are cut off by the truncation)
    assert int((want_r > 0.5).sum()) > int((want_r == 10.0).sum())


@pytest.mark.parametrize("B,rollout", [(131072, False), (262144, True)])
def test_big_state_only_batches_with_solution_plans(B, rollout):
    """State-only batches big enough for ONE LANE per environment (PW_OPT_STEP_LANE_BATCH defaults: 131 072 for one step per
    launch, 196 608 for pw_rollout): 64 steps, a third of the environments on their solution plans."""
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts, plans = _c3_pool()
    T, max_steps = 64, 40
    ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
    acts, driven = _actions(np.random.default_rng(B), ids, plans, T, 3)
    vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=max_steps, observation=None, device=0,
                       autoreset=True)
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    want_r, _, _ = _run_against_digest(vec, oracles, ids, acts, max_steps, rollout=rollout)
    assert int((want_r == 10.0).sum()) > 0 and driven >= B // 4
