"""-m gpu: the whole-grid 16 x 16 formulation of the step (step_quad16_body: four lanes per environment, every board in
registers, the puzzle in one 576-byte record) against the C oracle (puzzle.py:348-411, gym_env.py:201-226) and against the
lane-group kernels it replaces, on exactly the inputs that formulation has to be careful with: movables overlapping walls and
each other, movables partly or wholly OUTSIDE the grid (the bounds clause of puzzle.py:557-561: the quad leaves its boards and
asks the tables), launches of many steps in which a movable leaves the grid and comes back, next-step autoreset, N_pad 4 / 8 /
32 pools and workgroups that mix puzzles with and without a record.

(What the engine defines: every movable inside the grid, or up to one cell beyond it while still covering a grid cell -- where a
movable that overlaps a wall can walk to.  A movable ENTIRELY outside the grid, or further out, is outside that domain: every
formulation of the step -- rows, boards, tables -- lets the border wall stop it on its way back in, the reference's tables, which
hold in-bounds positions only, do not.  No such state is reachable without first overlapping the border wall)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _pool(kind):
    """(texts, label): Level-0 pools by N_pad, and a mixed pool with Level 1-3 puzzles between the Level-0 ones."""
    from pushworld_amd import benchmark_data as bd

    l0 = list(bd.level0_texts(limit=40).values())  # 7 families x 40
    n_of = [_num_movables(t) for t in l0]
    if kind == "np4":
        return [t for t, n in zip(l0, n_of) if n <= 4]
    if kind == "np8":
        five = [t for t, n in zip(bd.level0_texts(("all", "obstacles")).values(), map(_num_movables, bd.level0_texts(("all", "obstacles")).values())) if n == 5][:40]
        assert len(five) >= 10
        return l0[:120] + five
    texts = list(l0[:150])
    for lv in (1, 2):  # (Level 2 has the one puzzle with more than 16 movables, `Clean Sweep`: N_pad 32)
        for p in bd.level_paths(lv)[:30]:
            with open(p) as f:
                texts.append(f.read())
    return texts


def _num_movables(text):
    names = set()
    for cell in text.split():
        for e in cell.lower().split("+"):
            if e == "a" or e[0] == "m":
                names.add(e)
    return len(names)


def _random_states(rng, oracles, ids, np_pad):
    """int8 [B, np_pad, 2]: a third initial states with one movable displaced, a third uniformly inside the grid (overlaps of
    walls and movables included), a third partly outside the grid."""
    B = len(ids)
    pos = np.zeros((B, np_pad, 2), np.int8)
    for b, pid in enumerate(ids):
        o = oracles[pid]
        n = o.num_movables
        init = np.array(o.initial_state, np.int64)
        mode = b % 3
        if mode == 0:
            st = init.copy()
            j = rng.integers(0, n)
            st[j] += rng.integers(-1, 2, size=2)
        elif mode == 1:
            st = np.zeros((n, 2), np.int64)
            for j in range(n):
                w = 1 + max(c[0] for c in o.py.shapes[j]) - min(c[0] for c in o.py.shapes[j])
                hh = 1 + max(c[1] for c in o.py.shapes[j]) - min(c[1] for c in o.py.shapes[j])
                st[j] = [rng.integers(0, o.width - w + 1), rng.integers(0, o.height - hh + 1)]
        else:  # one cell beyond the grid at most and never ENTIRELY outside it (see the module docstring)
            st = np.zeros((n, 2), np.int64)
            for j in range(n):
                w = 1 + max(c[0] for c in o.py.shapes[j]) - min(c[0] for c in o.py.shapes[j])
                hh = 1 + max(c[1] for c in o.py.shapes[j]) - min(c[1] for c in o.py.shapes[j])
                while True:  # (some CELL of the movable inside the grid, not just its bounding box)
                    st[j] = [rng.integers(max(-1, 1 - w), min(o.width - w + 1, o.width - 1) + 1),
                             rng.integers(max(-1, 1 - hh), min(o.height - hh + 1, o.height - 1) + 1)]
                    if any(0 <= st[j][0] + cx < o.width and 0 <= st[j][1] + cy < o.height for cx, cy in o.py.shapes[j]):
                        break
        pos[b, :n] = st
    return pos


@pytest.mark.parametrize("kind", ["np4", "np8", "mixed"])
def test_single_steps_from_arbitrary_states_match_the_oracle(kind):
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts = _pool(kind)
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    reps = 24
    ids = np.repeat(np.arange(len(texts)), reps)  # sorted by puzzle: whole workgroups of one kind, mixed ones at the seams
    B = len(ids)
    pool = [PushWorldPuzzle(text=t) for t in texts]
    quad = VecPushWorld(pool, B, puzzle_ids=ids, observation=None, device=0, max_steps=1000)
    plain = VecPushWorld(pool, B, puzzle_ids=ids, observation=None, device=0, max_steps=1000, engine_options={"step_quad16": "never"})
    fits = quad.engine.get_option("step_quad16_puzzles")
    assert fits >= (len(texts) if kind != "mixed" else 150) and plain.engine.get_option("step_quad16") == 2
    NP = quad.num_objects_padded
    assert NP == {"np4": 4, "np8": 8, "mixed": 32}[kind]
    rng = np.random.default_rng(5)
    pos0 = _random_states(rng, oracles, ids, NP)
    quad.reset()
    plain.reset()
    for act in range(4):
        acts = torch.full((B,), act, dtype=torch.uint8, device=quad.device)
        for v in (quad, plain):
            v.set_states(pos0)
            v.steps.zero_()
            v.terminated.zero_()
            v.truncated.zero_()
        _, rq, tq, uq = quad.step(acts)
        _, rp, tp, up = plain.step(acts)
        got = quad.states()
        assert (got == plain.states()).all()
        assert (rq.cpu().numpy().view(np.uint64) == rp.cpu().numpy().view(np.uint64)).all()
        assert (tq.cpu().numpy() == tp.cpu().numpy()).all() and (uq.cpu().numpy() == 0).all()
        rq, tq = rq.cpu().numpy(), tq.cpu().numpy()
        for b in range(B):
            o = oracles[ids[b]]
            n = o.num_movables
            nxt, rew, term = o.env_step([tuple(int(v) for v in p) for p in pos0[b, :n]], act)
            assert (got[b, :n] == np.array(nxt, np.int64)).all(), (kind, b, act, pos0[b, :n], got[b, :n], nxt)
            assert (got[b, n:] == 0).all()
            assert np.float64(rew).view(np.uint64) == rq[b].view(np.uint64) and bool(tq[b]) == term, (kind, b, act)


@pytest.mark.parametrize("kind,autoreset", [("np4", True), ("np8", True), ("mixed", True), ("mixed", False)])
def test_rollouts_match_the_oracle(kind, autoreset):
    """96 steps in one launch from the initial states (next-step autoreset, max_steps 17): every step's reward / flags and the
    final state against the C oracle's trace."""
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts = _pool(kind)
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    ids = np.repeat(np.arange(len(texts)), 16)
    B, T = len(ids), 96
    vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, observation=None, device=0, max_steps=17,
                       autoreset=autoreset)
    rng = np.random.default_rng(11)
    acts = rng.integers(0, 4, size=(T, B), dtype=np.uint8)
    want_pos, want_r, want_te, want_tr, want_steps = c_oracle.rollout_trace(oracles, ids, acts, 17, autoreset, vec.num_objects_padded)
    vec.reset()
    rh, th, uh = vec.rollout(torch.as_tensor(acts).to(vec.device), history=True)
    assert (vec.states() == want_pos[-1]).all()
    assert (rh.cpu().numpy().view(np.uint64) == want_r.view(np.uint64)).all()
    assert (th.cpu().numpy() == want_te).all() and (uh.cpu().numpy() == want_tr).all()
    assert (vec.steps.cpu().numpy() == want_steps[-1]).all()
    c = vec.counters()
    assert c["env_steps"] == B * T and c["episodes_ended"] == int(((want_te | want_tr) != 0).sum())


def test_rollouts_through_the_border_equal_single_steps():
    """Launches of several steps from states whose movables overlap the border walls: they walk out of the grid and back in
    during the launch (boards -> tables -> boards).  pw_rollout on the 16 x 16 boards == the same steps one launch at a time
    on the lane-group kernels (which the test above pins to the oracle step by step)."""
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts = _pool("np8")
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    ids = np.repeat(np.arange(len(texts)), 16)
    B, T = len(ids), 24
    pool = [PushWorldPuzzle(text=t) for t in texts]
    quad = VecPushWorld(pool, B, puzzle_ids=ids, observation=None, device=0, max_steps=9, autoreset=True)
    plain = VecPushWorld(pool, B, puzzle_ids=ids, observation=None, device=0, max_steps=9, autoreset=True,
                         engine_options={"step_quad16": "never"})
    rng = np.random.default_rng(3)
    pos0 = np.zeros((B, quad.num_objects_padded, 2), np.int8)
    for b, pid in enumerate(ids):
        o = oracles[pid]
        n = o.num_movables
        st = np.array(o.initial_state, np.int64)
        for j in rng.choice(n, size=min(n, 2), replace=False):  # onto the border ring (walls: it overlaps them and may leave)
            side = rng.integers(0, 4)
            st[j] = [(0, st[j][1]), (o.width - 1, st[j][1]), (st[j][0], 0), (st[j][0], o.height - 1)][side]
        pos0[b, :n] = st
    # two thirds of the steps push towards one side: the displaced movables leave the grid within the launch
    side = rng.integers(0, 4, size=B)
    acts = np.where(rng.random((T, B)) < 0.66, side[None, :], rng.integers(0, 4, size=(T, B))).astype(np.uint8)
    single = VecPushWorld(pool, B, puzzle_ids=ids, observation=None, device=0, max_steps=9, autoreset=True)
    quad.reset()
    plain.reset()
    single.reset()
    quad.set_states(pos0)
    plain.set_states(pos0)
    single.set_states(pos0)
    acts_dev = torch.as_tensor(acts).to(quad.device)
    rh, th, uh = quad.rollout(acts_dev, history=True)
    left = 0
    for t in range(T):
        _, r, te, tr = plain.step(acts_dev[t])
        _, r1, te1, tr1 = single.step(acts_dev[t])  # one step per launch: the boards, the tables while a movable is outside
        assert (single.states() == plain.states()).all(), t
        assert (r1.cpu().numpy().view(np.uint64) == r.cpu().numpy().view(np.uint64)).all() and (te1.cpu().numpy() == te.cpu().numpy()).all(), t
        assert (rh[t].cpu().numpy().view(np.uint64) == r.cpu().numpy().view(np.uint64)).all(), t
        assert (th[t].cpu().numpy() == te.cpu().numpy()).all() and (uh[t].cpu().numpy() == tr.cpu().numpy()).all(), t
        p = plain.states()
        left += int(((p[:, :, 0] < 0) | (p[:, :, 1] < 0)).any(axis=1).sum())
    assert (quad.states() == plain.states()).all()
    assert (quad.steps.cpu().numpy() == plain.steps.cpu().numpy()).all()
    assert left > B // 8  # the launch did see states with a movable outside the grid
