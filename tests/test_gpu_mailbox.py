"""-m gpu: the resident step kernel (pw_mailbox_*, K1f) against pw_step on a twin batch, step by step and bit for bit:
positions, step counters, float64 rewards, terminated / truncated (with next-step autoreset, truncation, solved episodes from the
puzzles' solution-like random play, and actions outside 0..3), the pinned host verdicts of every step, posts that run ahead of the
waits, host and device actions, the counters after the close, the idle limit, and the refusals while a mailbox is open."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _pool(n):
    from pushworld_amd import benchmark_data as bd

    texts = [t for t in bd.level0_texts().values()]
    small = []
    for t in texts:
        rows = [r for r in t.strip().splitlines()]
        if len(rows) + 2 <= 8 and len(rows[0].split()) + 2 <= 8:
            small.append(t)
        if len(small) == n:
            break
    assert len(small) == n
    return small


def _twins(B, n_puzzles, max_steps, autoreset=True):
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    texts = _pool(n_puzzles)
    ids = (np.arange(B, dtype=np.int64) * n_puzzles) // B
    mk = lambda: VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=max_steps, observation=None,
                              device=0, autoreset=autoreset)
    a, b = mk(), mk()
    assert a.engine.get_option("step_board_set") == 1
    a.reset()
    b.reset()
    return a, b


def _same_state(a, b):
    import torch

    torch.cuda.synchronize()
    assert (a.pos == b.pos).all() and (a.steps == b.steps).all()
    assert (a.terminated == b.terminated).all() and (a.truncated == b.truncated).all()
    assert (a.reward.view(torch.int64) == b.reward.view(torch.int64)).all()


@pytest.mark.parametrize("B,host_actions,mode", [(4096, True, 2), (4096, False, 2), (1000, True, 2), (37, False, 2), (20000, True, 2),
                                                  (4096, True, 0), (3000, False, 1), (4096, False, 6), (700, True, 4), (5000, True, 5), (4096, True, 3), (2500, False, 3), (4096, False, 7),
                                                  (4096, True, 11), (4096, False, 11), (333, False, 11)])
def test_mailbox_steps_equal_pw_step(B, host_actions, mode):
    import torch

    T = 160
    a, b = _twins(B, 23, max_steps=25)
    assert a.engine.get_option("mailbox_mode") == 11  # (3 pipelined: the default since round 6)
    a.engine.set_option("mailbox_mode", mode)  # who polls the host's word, fences or system-scope accesses: same results
    rng = np.random.default_rng(B)
    acts = rng.integers(0, 4, size=(T, B), dtype=np.uint8)
    acts[40, ::7] = 9      # outside Discrete(4)
    acts[41, 3::11] = 255
    acts_dev = torch.as_tensor(acts).to(a.device)
    torch.cuda.synchronize()
    ended = solved = 0
    with a.mailbox() as mb:
        for t in range(T):
            r, te, tr = mb.step(acts[t] if host_actions else acts_dev[t])
            _, r2, te2, tr2 = b.step(acts_dev[t])
            assert (r.view(np.uint64) == r2.cpu().numpy().view(np.uint64)).all(), t
            assert (te == te2.cpu().numpy()).all() and (tr == tr2.cpu().numpy()).all(), t
            ok = te != 0xFF
            ended += int(((te | tr) != 0)[ok].sum())
            solved += int((te != 0)[ok].sum())
            if t % 20 == 19 or t in (40, 41, 42):  # the device arrays after this step (the kernel is still resident)
                assert (a.pos.cpu() == b.pos.cpu()).all() and (a.steps.cpu() == b.steps.cpu()).all(), t
        with pytest.raises(ValueError):
            a.step(acts_dev[0])  # the environments live in the resident kernel
        with pytest.raises(ValueError):
            a.reset()
    _same_state(a, b)
    ca, cb = a.counters(), b.counters()
    assert ca == cb and ca["env_steps"] == B * T and ca["episodes_ended"] == ended and ca["episodes_solved"] == solved
    # (actions outside 0..3 on the reset step of an episode are ignored, not counted)
    assert 0 < ca["bad_actions"] <= len(acts[40, ::7]) + len(acts[41, 3::11]) and ended > B  # (truncations at 25 steps: several per env)
    # ... and the batch carries on through ordinary launches
    a.step(acts_dev[5])
    b.step(acts_dev[5])
    _same_state(a, b)


def test_posts_ahead_of_the_waits_and_reopen():
    import torch

    B, T = 4096, 96
    a, b = _twins(B, 5, max_steps=30)
    acts = np.random.default_rng(3).integers(0, 4, size=(T, B), dtype=np.uint8)
    acts_dev = torch.as_tensor(acts).to(a.device)
    torch.cuda.synchronize()
    want = []
    for t in range(T):
        _, r, te, tr = b.step(acts_dev[t])
        want.append((r.cpu().numpy().copy(), te.cpu().numpy().copy(), tr.cpu().numpy().copy()))
    for first, ring in ((0, 8), (T // 2, 2)):  # two mailboxes in turn over the same batch, different rings
        with a.engine.mailbox(a.puzzle_id, a.pos, a.steps, a.reward, a.dgoals, a.terminated, a.truncated, a.flags, ring, 1000) as mb:
            pending = []
            for t in range(first, first + T // 2):
                pending.append((t, mb.post(acts_dev[t])))  # (beyond `ring` posts ahead post() itself waits)
                if len(pending) == ring:
                    tt, seq = pending.pop(0)
                    r, te, tr = mb.wait(seq)
                    assert (r.view(np.uint64) == want[tt][0].view(np.uint64)).all() and (te == want[tt][1]).all() and (tr == want[tt][2]).all(), tt
            for tt, seq in pending:
                r, te, tr = mb.wait(seq)
                assert (r.view(np.uint64) == want[tt][0].view(np.uint64)).all() and (te == want[tt][1]).all() and (tr == want[tt][2]).all(), tt
    _same_state(a, b)


@pytest.mark.parametrize("ring,ahead,host", [(2, 2, False), (2, 1, True), (4, 4, True), (8, 8, False), (64, 64, False)])
def test_soak_slots_come_round(ring, ahead, host):
    """30 000 steps through pw_mailbox_run with every slot of the ring (the host's words, the relay words, the arrival counters,
    the result slots) reused thousands of times at the shortest distance the protocol allows: the final state, the counters and the
    verdicts of the last step equal those of pw_rollout over the same actions on a twin."""
    import torch

    B, T = 4096, 30000
    a, b = _twins(B, 11, max_steps=40)
    acts = np.random.default_rng(ring * 100 + ahead).integers(0, 4, size=(T, B), dtype=np.uint8)
    acts_dev = torch.as_tensor(acts).to(a.device)
    torch.cuda.synchronize()
    for k in range(0, T, 1000):
        b.rollout(acts_dev[k:k + 1000])
    mb = a.engine.mailbox(a.puzzle_id, a.pos, a.steps, a.reward, a.dgoals, a.terminated, a.truncated, a.flags, ring, 2000)
    last = mb.run(acts if host else acts_dev, ahead)
    assert last == T
    r, te, tr = mb.wait(last)
    assert (r.view(np.uint64) == b.reward.cpu().numpy().view(np.uint64)).all()
    assert (te == b.terminated.cpu().numpy()).all() and (tr == b.truncated.cpu().numpy()).all()
    prof = mb.close(profile=True)
    assert prof["steps"] == T and prof["ended_by"] == "stop"
    _same_state(a, b)
    assert a.counters() == b.counters()


@pytest.mark.parametrize("B,ahead", [(65536, 8), (65536 - 100, 1), (40000, 3)])
def test_largest_batches_in_flight(B, ahead):
    """The most environments a mailbox takes (65 536 = 1 024 stepping wavefronts + the relay and the publisher workgroup, all resident):
    every wavefront's own step count, the publisher's look over all of them, posts ahead of the waits -- the state, the counters and the
    last verdicts after 1 500 steps equal pw_rollout's on a twin."""
    import torch

    T = 1500
    a, b = _twins(B, 29, max_steps=30)
    acts = np.random.default_rng(B + ahead).integers(0, 4, size=(T, B), dtype=np.uint8)
    acts_dev = torch.as_tensor(acts).to(a.device)
    torch.cuda.synchronize()
    for k in range(0, T, 500):
        b.rollout(acts_dev[k:k + 500])
    with a.mailbox(ring=8) as mb:
        last = mb.run(acts_dev, ahead)
        assert last == T
        r, te, tr = mb.wait(last)
        assert (r.view(np.uint64) == b.reward.cpu().numpy().view(np.uint64)).all()
        assert (te == b.terminated.cpu().numpy()).all() and (tr == b.truncated.cpu().numpy()).all()
    _same_state(a, b)
    assert a.counters() == b.counters()


def test_idle_limit_ends_the_kernel():
    import torch

    B = 512
    a, b = _twins(B, 3, max_steps=50)
    acts = np.random.default_rng(5).integers(0, 4, size=(4, B), dtype=np.uint8)
    mb = a.mailbox(idle_ms=200)
    mb.step(acts[0])
    mb.step(acts[1])
    time.sleep(0.45)
    t0 = time.time()
    torch.cuda.synchronize()  # the kernel has ended by itself: a device-wide synchronisation returns
    assert time.time() - t0 < 1.0
    # ... and the next post finds the mailbox expired with no step in flight: it is opened again over the same arrays (they hold
    # the state after the last complete step) and the step is posted to the new kernel
    seq = mb.post(acts[2])
    assert mb.reopened == 1 and seq == 1
    mb.wait(seq)
    for t in range(3):
        b.step(torch.as_tensor(acts[t]).to(b.device))
    assert (a.pos.cpu() == b.pos.cpu()).all() and (a.steps.cpu() == b.steps.cpu()).all()
    # with a step in flight (posted, not waited for) the caller still holds a number of the old kernel: the error is raised
    mb.post(acts[3])
    time.sleep(0.45)
    with pytest.raises(RuntimeError):
        mb.post(acts[0])
    mb.close()
    b.step(torch.as_tensor(acts[3]).to(b.device))
    _same_state(a, b)  # the arrays hold the state after the last complete step
    with a.mailbox() as mb2:  # a new one carries on
        mb2.step(acts[2])
    b.step(torch.as_tensor(acts[2]).to(b.device))
    _same_state(a, b)


@pytest.mark.parametrize("pool,B,tables", [("level1", 4096, True), ("level1", 1500, False), ("c4mix", 4096, True), ("level0_big", 3000, True),
                                           ("heavy", 1024, True)])
def test_any_set_through_lane_step(pool, B, tables):
    """Sets that do not fit 8 x 8 boards: the resident kernel steps them with lane_step (one lane per environment; overlap tables
    where the engine has them for every puzzle, row bitboards otherwise) -- N_pad 16 (Level 1), 32 (the C4 mix with `Clean Sweep`),
    8 (Level-0 puzzles beyond 8 x 8 cells) -- against pw_step on a twin, a third of the environments on their solution plans."""
    import torch

    import bench
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld
    from test_gpu_deep import _actions, _plan

    plans = {}
    if pool == "level1":
        paths = bench.level1_paths()
        texts = [open(p).read() for p in paths]
        plans = {i: _plan("level1", p) for i, p in enumerate(paths)}
    elif pool == "c4mix":
        texts = list(bd.level0_texts().values())[:40]
        for lv in (1, 2, 3, 4):
            for p in bd.level_paths(lv)[:12]:
                plans[len(texts)] = _plan(f"level{lv}", p)
                texts.append(open(p).read())
        texts.append(open([p for p in bd.level_paths(2) if "Clean Sweep" in p][0]).read())
    elif pool == "heavy":  # 12 .. 19 movables: four and eight lanes per environment in the segment form
        texts = []
        for lv, name in ((2, "Clean Sweep"), (4, "Mind The Gap"), (3, "Yin Yang"), (3, "Rocky Shore")):
            p = [q for q in bd.level_paths(lv) if name in q][0]
            plans[len(texts)] = _plan(f"level{lv}", p)
            texts.append(open(p).read())
    else:
        texts = [t for t in bd.level0_texts().values() if len(t.strip().splitlines()) + 2 > 8][:30]
        assert len(texts) == 30
    T, max_steps = 120, 60
    ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
    acts, _ = _actions(np.random.default_rng(B), ids, plans, T, 3)
    acts[50, ::5] = 77
    mk = lambda: VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=max_steps, observation=None,
                              device=0, autoreset=True, engine_options=None if tables else {"step_tables": "none"})
    a, b = mk(), mk()
    assert a.engine.get_option("step_board_set") == 0
    a.reset()
    b.reset()
    acts_dev = torch.as_tensor(acts).to(a.device)
    torch.cuda.synchronize()
    solved = 0
    with a.mailbox() as mb:
        # (round 6: a batch whose every environment sits in a segment of its binding is stepped by the segments -- Level 1 at 60
        # environments per puzzle, the big Level-0 puzzles at 100; the others one lane per environment)
        assert a.engine.get_option("mailbox_form") == (2 if (pool, B) in (("level1", 4096), ("level0_big", 3000), ("heavy", 1024)) and tables else 1)
        for t in range(T):
            r, te, tr = mb.step(acts[t] if t % 2 else acts_dev[t])
            _, r2, te2, tr2 = b.step(acts_dev[t])
            assert (r.view(np.uint64) == r2.cpu().numpy().view(np.uint64)).all(), t
            assert (te == te2.cpu().numpy()).all() and (tr == tr2.cpu().numpy()).all(), t
            solved += int((r == 10.0).sum())
            if t % 30 == 29 or t == 50:
                assert (a.pos.cpu() == b.pos.cpu()).all() and (a.steps.cpu() == b.steps.cpu()).all(), t
    _same_state(a, b)
    assert a.counters() == b.counters()
    if plans:
        assert solved > 0


def test_refusals():
    a, _ = _twins(64, 2, max_steps=10)
    with a.mailbox() as mb:
        with pytest.raises(ValueError):
            a.mailbox()
        with pytest.raises(ValueError):
            mb.post(np.zeros(63, np.uint8))
        with pytest.raises(ValueError):
            mb.wait(5)
        with pytest.raises(ValueError):
            a.rollout(__import__("torch").zeros((2, 64), dtype=__import__("torch").uint8, device=a.device))


# ---------------------------------------------------------------------------------------------------------------------------------
# The resident kernel against the ORACLE: the tests above compare the mailbox with pw_step on a twin batch (HIP vs
# HIP); here every posted step is checked against oracle/pw_oracle.c (puzzle.py:348-411, gym_env.py:201-226) -- the 64-bit digest
# of every position row, float64 reward bits, terminated, truncated, the step counters -- read from the PINNED result slots and
# from the device arrays the kernel keeps up to date.
# ---------------------------------------------------------------------------------------------------------------------------------
def _bfs_plan(oracle, cap=200000):
    """A shortest plan of a small puzzle by breadth-first search over the C oracle's get_next_state / goal test."""
    start = tuple(oracle.initial_state)
    seen = {start: None}
    frontier = [start]
    while frontier and len(seen) < cap:
        nxt = []
        for s in frontier:
            for a in range(4):
                t, _, term = oracle.env_step(s, a)
                if t in seen:
                    continue
                seen[t] = (s, a)
                if term:
                    plan = []
                    while seen[t] is not None:
                        t, a2 = seen[t]
                        plan.append(a2)
                    return np.array(plan[::-1], np.uint8)
                nxt.append(t)
        frontier = nxt
    return None


def _oracle_actions(rng, ids, plans, T, every):
    B = len(ids)
    acts = rng.integers(0, 4, size=(T, B), dtype=np.uint8)
    driven = 0
    for b in range(0, B, every):
        plan = plans.get(int(ids[b]))
        if plan is None or len(plan) == 0:
            continue
        cyc = np.concatenate([plan, np.zeros(1, np.uint8)])  # (one ignored action: the step on which the solved episode is reset)
        acts[:, b] = cyc[np.arange(T) % len(cyc)]
        driven += 1
    return acts, driven


def _mailbox_against_oracle(texts, plans, ids, T, max_steps, host_actions, ahead, seed, seg=None):
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    B = len(ids)
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    acts, driven = _oracle_actions(np.random.default_rng(seed), ids, plans, T, 3)
    assert driven >= B // 4
    vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=max_steps, observation=None, device=0,
                       autoreset=True)
    NP = vec.num_objects_padded
    w = np.random.default_rng(7).integers(-2**62, 2**62, size=NP * 2, dtype=np.int64)
    want_d, want_r, want_te, want_tr, want_steps, want_last = c_oracle.rollout_digest(oracles, ids, acts, max_steps, True, NP, w)
    w_dev = torch.as_tensor(w).to(vec.device)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    vec.reset()
    torch.cuda.synchronize()
    if seg is not None:  # the form of the resident kernel: the segments of the bound batch (True) or one lane per environment (False)
        vec.engine.set_option("mailbox_seg", 0 if seg else 2)

    def check_slot(t, slot):
        r, te, tr = slot
        assert (r.view(np.uint64) == want_r[t].view(np.uint64)).all(), t
        assert (te == want_te[t]).all() and (tr == want_tr[t]).all(), t

    def check_device(t):  # (the kernel is resident: the arrays are read by copies on torch's stream after the step completed)
        d = (vec.pos.view(B, NP * 2).to(torch.int64) * w_dev).sum(dim=1).cpu().numpy()
        bad = np.nonzero(d != want_d[t])[0]
        assert bad.size == 0, (t, bad[:5])
        assert (vec.steps.cpu().numpy() == want_steps[t]).all(), t
        assert (vec.reward.cpu().numpy().view(np.uint64) == want_r[t].view(np.uint64)).all(), t
        assert (vec.terminated.cpu().numpy() == want_te[t]).all() and (vec.truncated.cpu().numpy() == want_tr[t]).all(), t

    with vec.mailbox(ring=max(8, ahead)) as mb:
        if seg is not None:
            assert vec.engine.get_option("mailbox_form") == (2 if seg else 1)
        if ahead <= 1:
            for t in range(T):
                check_slot(t, mb.step(acts[t] if host_actions else acts_dev[t]))
                check_device(t)
        else:
            pending = []
            for t in range(T):
                pending.append((t, mb.post(acts[t] if host_actions else acts_dev[t])))
                if len(pending) == ahead:
                    tt, seq = pending.pop(0)
                    check_slot(tt, mb.wait(seq))
            for tt, seq in pending:
                check_slot(tt, mb.wait(seq))
            check_device(T - 1)
    torch.cuda.synchronize()
    assert (vec.states() == want_last).all()
    assert (vec.steps.cpu().numpy() == want_steps[-1]).all()
    solved = int((want_r == 10.0).sum())
    assert solved >= driven and int(((want_te | want_tr) != 0).sum()) > B  # solved episodes, truncations: autoreset inside the loop
    c = vec.counters()
    assert c["env_steps"] == B * T and c["episodes_solved"] == int((want_te != 0).sum())
    assert c["episodes_ended"] == int(((want_te | want_tr) != 0).sum()) and c["bad_actions"] == 0


@pytest.mark.parametrize("host_actions,ahead", [(True, 1), (False, 1), (False, 8)])
def test_c2_mailbox_against_the_oracle(host_actions, ahead):
    """BASELINE config 2 (4 096 copies of level0/base/train/level_0_base_train_0) through the resident kernel's board formulation."""
    from oracle import c_oracle
    from pushworld_amd import benchmark_data as bd

    text = next(iter(bd.level0_texts(("base",), "train", 1).values()))
    plan = _bfs_plan(c_oracle.COraclePuzzle(text))
    assert plan is not None and len(plan) > 0
    ids = np.zeros(4096, np.int64)
    _mailbox_against_oracle([text], {0: plan}, ids, 416, 25, host_actions, ahead, seed=11 + ahead)


@pytest.mark.parametrize("host_actions,ahead,seg", [(False, 1, True), (True, 8, True), (False, 8, True), (False, 1, False), (True, 8, False)])
def test_level1_mailbox_against_the_oracle(host_actions, ahead, seg):
    """4 096 environments over the 68 Level-1 puzzles (the resident kernel's one-lane-per-environment table formulation), a third of
    them on the human solution plans of data/solutions."""
    import bench

    paths = bench.level1_paths()
    texts = [open(p).read() for p in paths]
    sol = os.path.join(ROOT, "pushworld_amd", "data", "solutions", "level1")
    plans = {}
    for i, p in enumerate(paths):
        with open(os.path.join(sol, os.path.splitext(os.path.basename(p))[0] + ".yaml")) as f:
            for line in f:
                if line.startswith("plan:"):
                    plans[i] = np.array(["LRUD".index(c) for c in line.split(":", 1)[1].strip()], np.uint8)
    B = 4096
    ids = (np.arange(B, dtype=np.int64) * len(texts)) // B
    _mailbox_against_oracle(texts, plans, ids, 400, 120, host_actions, ahead, seed=23 + ahead, seg=seg)
