"""GPU breadth-first frontier services (pw_search_*) against a sequential FIFO
breadth-first search over the oracle: identical state numbering, links, layers and plans."""
from collections import deque

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def host_bfs(oz, start=None, max_states=None):
    """FIFO BFS, actions 0..3; returns states (list of tuples), parent, action, layer sizes, first goal."""
    s0 = tuple(start) if start is not None else oz.initial_state
    is_goal = oz.py.is_goal_state  # the compiled oracle steps, the Python oracle tests the goal
    states, parent, action, depth = [s0], [-1], [255], [0]
    index = {s0: 0}
    goal = 0 if is_goal(s0) else -1
    q = deque([0])
    while q:
        i = q.popleft()
        for a in range(4):
            n = oz.get_next_state(states[i], a)
            if n == states[i] or n in index:
                continue
            index[n] = len(states)
            states.append(n)
            parent.append(i)
            action.append(a)
            depth.append(depth[i] + 1)
            if goal < 0 and is_goal(n):
                goal = len(states) - 1
            q.append(len(states) - 1)
            if max_states is not None and len(states) >= max_states:
                return states, parent, action, depth, goal
    return states, parent, action, depth, goal


CASES = ["pytest:trivial.pwp", "pytest:trivial_obstacle.pwp", "pytest:trivial_tool.pwp", "pytest:pushing.pwp",
         "pytest:transitive_pushing.pwp", "l0:level0/base/train/level_0_base_train_0.pwp",
         "l0:level0/all/train/level_0_all_train_3.pwp", "rand:3", "rand:17", "rand:42"]


@pytest.mark.parametrize("chunk", [None, "7", "lanes", "mixed", "keys", "keys7", "keyslanes"])
def test_bfs_numbering_equals_sequential_search(golden, chunk, monkeypatch):
    """Whole reachable space (capped at 60 000 states): every state, parent, action, layer boundary
    and the first goal index equal the host FIFO search; chunk=7 forces many passes per layer.  "lanes": the
    one-lane-per-parent expansion kernel (what passes of >= 131 072 parents run by themselves) for every pass; "mixed":
    lane groups and lanes alternating layer by layer (both must hash a state to the same slot)."""
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch

    # "keys": published entries of the closed set are the exact packed states wherever a state fits 63 bits (PW_OPT_SEARCH_KEYS
    # exact: every puzzle of CASES but the widest); the default is fingerprint + index entries for every puzzle
    keys = bool(chunk) and chunk.startswith("keys")
    if keys:
        chunk = chunk[4:] or None
    lanes = chunk in ("lanes", "mixed")
    mixed = chunk == "mixed"
    chunk = None if lanes else chunk
    chunk_arg = int(chunk) if chunk else None
    cap = 3000 if chunk else 60000
    n_checked = 0
    for key in CASES:
        if key not in golden.meta:
            continue
        text = golden.text(key)
        oz = c_oracle.COraclePuzzle(text)
        want_states, want_parent, want_action, want_depth, want_goal = host_bfs(oz, max_states=cap + 1)
        if len(want_states) > cap:
            continue  # space larger than the cap: covered by the truncated test below
        pz = PushWorldPuzzle(text=text)
        pz._engine().set_option("search_keys", "exact" if keys else "fingerprint")
        bfs = BreadthFirstSearch(pz, max_states=cap + 8, chunk=chunk_arg)
        bfs.begin()
        layer = 0
        while not bfs.exhausted:
            if lanes:
                pz._engine().set_option("step_kernel", "group" if (mixed and layer % 2) else "lane")
            bfs.expand()
            layer += 1
        assert bfs.total_states == len(want_states), key
        got = bfs.states()
        assert (got == np.array(want_states, dtype=np.int64).reshape(got.shape)).all(), key
        par, act = bfs.links()
        assert (par == np.array(want_parent)).all() and (act[1:] == np.array(want_action[1:])).all(), key
        sizes = np.bincount(np.array(want_depth))
        assert [c for _, c in bfs.layers] == sizes.tolist(), key
        assert bfs.goal_index == want_goal, key
        if want_goal >= 0:
            plan = bfs.plan(want_goal)
            assert len(plan) == want_depth[want_goal]
            assert pz.is_valid_plan(plan)
        bfs.close()
        n_checked += 1
    assert n_checked >= 5


def test_bfs_solves_benchmark_puzzles_optimally(golden):
    """Level-1 puzzles with a small reachable space: solve() returns a valid plan no longer than the
    human solution, equal in length to the host BFS optimum; start states other than the initial one."""
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch

    solved = 0
    for key in [k for k in golden.keys if k.startswith("l0:")][::60][:8] + ["pytest:trivial_tool.pwp"]:
        text = golden.text(key)
        pz = PushWorldPuzzle(text=text)
        bfs = BreadthFirstSearch(pz, max_states=400000)
        try:
            plan = bfs.solve()
        except ValueError:
            bfs.close()
            continue
        oz = c_oracle.COraclePuzzle(text)
        _, _, _, want_depth, want_goal = host_bfs(oz, max_states=400000)
        if want_goal < 0:
            assert plan is None
        else:
            assert plan is not None and len(plan) == want_depth[want_goal] and pz.is_valid_plan(plan), key
            solved += 1
            # restart from the state after the first plan action: the rest of the plan is still optimal
            mid = pz.get_next_state(pz.initial_state, plan[0])
            bfs.begin(mid)
            rest = bfs.solve()
            assert rest is not None and len(rest) == len(plan) - 1
            s = mid
            for a in rest:
                s = pz.get_next_state(s, a)
            assert pz.is_goal_state(s)
        bfs.close()
    assert solved >= 4


def test_bfs_store_overflow_and_errors(golden):
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch

    pz = PushWorldPuzzle(text=golden.text("bench:level1/2 Obstacle.pwp"))
    bfs = BreadthFirstSearch(pz, max_states=500)
    with pytest.raises(ValueError):
        bfs.expand()  # begin() not called
    bfs.begin()
    with pytest.raises(ValueError):
        while True:
            bfs.expand()  # store full
    with pytest.raises(ValueError):
        bfs.expand()  # stays refused
    bfs.begin()  # a new search on the same object works again
    info = bfs.expand()
    assert info.depth == 1 and 1 <= info.new_states <= 4 and info.total_states == 1 + info.new_states
    with pytest.raises(ValueError):
        bfs.begin([(0, 0)])  # wrong arity
    with pytest.raises(ValueError):
        bfs.begin([(99, 0)] * pz.num_movables)  # outside the puzzle
    with pytest.raises(ValueError):
        bfs.states(0, 10 ** 9)
    with pytest.raises(ValueError):
        BreadthFirstSearch(pz, max_states=0)
    bfs.close()


@pytest.mark.parametrize("keys", ["fingerprint", "exact"])
@pytest.mark.parametrize("key,cap", [("bench:level1/2 Obstacle.pwp", 150000), ("bench:level2/Clean Sweep.pwp", 40000),
                                     ("bench:level2/Pull Dont Push.pwp", 300000)])
def test_bfs_large_layer_matches_host_prefix(golden, key, cap, keys):
    """'2 Obstacle' explored to 150 000 states with the default pass size: the first 150 000 states of
    the sequential search, in the same order (exercises multi-block scans and hash-table growth);
    'Clean Sweep' has 19 movables (32-lane groups, 10 words per state)."""
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch

    text = golden.text(key)
    oz = c_oracle.COraclePuzzle(text)
    pz = PushWorldPuzzle(text=text)
    pz._engine().set_option("search_keys", keys)  # exact 63-bit keys (`2 Obstacle`, `Pull Dont Push`; not the 19 movables of `Clean Sweep`)
    bfs = BreadthFirstSearch(pz, max_states=cap)
    bfs.begin()
    try:
        while not bfs.exhausted and bfs.goal_index < 0:
            bfs.expand()
    except ValueError:
        pass
    n_full = bfs.layers[-1][0] + bfs.layers[-1][1] if bfs.total_states < cap else bfs.layers[-1][0]
    # complete layers only (an overflowing layer is incomplete but still in sequential order)
    want_states, want_parent, want_action, _, _ = host_bfs(oz, max_states=cap)
    got = bfs.states(0, bfs.total_states)
    n = min(bfs.total_states, len(want_states))
    assert n >= min(cap, 20000) and n_full > 0
    assert (got[:n] == np.array(want_states[:n], dtype=np.int64).reshape(n, -1, 2)).all()
    par, act = bfs.links(0, n)
    assert (par == np.array(want_parent[:n])).all() and (act[1:] == np.array(want_action[1:n])).all()
    bfs.close()


# ------------------------------------------------------------------------------------------- novelty
def test_novelty_tables_reference_known_answers():
    """cpp/test/heuristics/test_novelty_heuristic.cc:86-106: the reference's own sequence, evaluated in
    one batch and again split over several calls (the tables persist between calls)."""
    import torch
    from pushworld_amd.search import NoveltyTables

    states = [[1, 2, 3, 4], [2, 3, 4, 5], [1, 3, 4, 5], [2, 3, 3, 5], [1, 3, 3, 5], [1, 3, 3, 4], [1, 3, 5, 4], [1, 3, 5, 4]]
    moved = [0b1111, 0b1111, 0b0001, 0b0100, 0b0101, 0b1000, 0b0100, 0]
    want = [1, 1, 2, 2, 3, 2, 1, 3]
    nt = NoveltyTables(4, 1, 6)
    st = torch.tensor(states, dtype=torch.int32, device=nt.device)
    mv = torch.tensor(moved, dtype=torch.int32, device=nt.device)
    assert nt.evaluate(st, mv).cpu().tolist() == want
    nt.reset()
    got = []
    for lo, hi in ((0, 1), (1, 4), (4, 8)):
        got += nt.evaluate(st[lo:hi].contiguous(), mv[lo:hi].contiguous()).cpu().tolist()
    assert got == want
    with pytest.raises(ValueError):  # y = 6 is outside the 1 x 6 grid
        nt.evaluate(torch.tensor([[1, 2, 6, 4]], dtype=torch.int32, device=nt.device), mv[:1].contiguous())
    with pytest.raises(ValueError):
        nt.evaluate(st[:, :3].contiguous(), mv)
    nt.close()


@pytest.mark.parametrize("n,w,h", [(5, 6, 5), (1, 3, 3), (12, 9, 7), (32, 4, 3)])
def test_novelty_tables_equal_sequential_oracle(n, w, h):
    """Random states and moved masks: the batched tables return exactly what the restated
    NoveltyHeuristic returns when fed the same states one at a time."""
    import torch
    from oracle import pw_oracle
    from pushworld_amd.search import NoveltyTables

    rng = np.random.default_rng(n * 100 + w)
    F = 6000
    xs, ys = rng.integers(0, w, (F, n)), rng.integers(0, h, (F, n))
    masks = rng.integers(0, 1 << min(n, 31), F).astype(np.int64)
    masks[rng.random(F) < 0.5] &= rng.integers(0, 1 << min(n, 31), F)[rng.random(F) < 0.5].sum() | 1  # sparser masks
    masks[::17] = 0
    if n == 32:
        masks[::5] |= 1 << 31
    oracle = pw_oracle.OracleNovelty(n)
    want = [oracle.estimate(tuple(zip(xs[k], ys[k])), [i for i in range(n) if (int(masks[k]) >> i) & 1]) for k in range(F)]
    nt = NoveltyTables(n, w, h)
    st = torch.as_tensor((xs * 10000 + ys).astype(np.int32)).to(nt.device)
    mv = torch.as_tensor(masks.astype(np.uint32).view(np.int32)).to(nt.device)
    got = []
    for lo, hi in ((0, 1), (1, 1000), (1000, 1003), (1003, F)):
        got += nt.evaluate(st[lo:hi].contiguous(), mv[lo:hi].contiguous()).cpu().tolist()
    assert got == want
    assert set(want) == ({1, 2, 3} if n > 1 else {1, 3})
    nt.close()


def host_iw(oz, width, max_states=None):
    """FIFO search with novelty pruning, the host model of pw_search_create(novelty_width = width)."""
    from oracle import pw_oracle

    s0 = oz.initial_state
    is_goal = oz.py.is_goal_state
    n = len(s0)
    nov = pw_oracle.OracleNovelty(n)
    nov.estimate(s0, range(n))
    states, parent, action, depth, pruned = [s0], [-1], [255], [0], [False]
    index = {s0: 0}
    goal = 0 if is_goal(s0) else -1
    q = deque([0])
    while q:
        i = q.popleft()
        for a in range(4):
            nxt, moved = oz.get_next_state_moved(states[i], a)
            if not moved or nxt in index:
                continue
            k = len(states)
            index[nxt] = k
            states.append(nxt)
            parent.append(i)
            action.append(a)
            depth.append(depth[i] + 1)
            if goal < 0 and is_goal(nxt):
                goal = k
            cut = nov.estimate(nxt, moved) > width
            pruned.append(cut)
            if not cut:
                q.append(k)
            if max_states is not None and len(states) >= max_states:
                return states, parent, action, depth, pruned, goal
    return states, parent, action, depth, pruned, goal


@pytest.mark.parametrize("width", [1, 2])
@pytest.mark.parametrize("chunk", [None, "5", "lanes"])
def test_width_limited_search_equals_host_model(golden, width, chunk, monkeypatch):
    """IW(1) / IW(2): states, links, pruned flags, layers and first goal equal the host model built from
    the oracle step and the restated NoveltyHeuristic; with 5-parent passes as well, and with the one-lane-per-parent
    expansion kernel ("lanes")."""
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch

    lanes = chunk == "lanes"
    chunk = None if lanes else chunk
    chunk_arg = int(chunk) if chunk else None
    keys = CASES + ["bench:level1/2 Obstacle.pwp", "bench:level1/Choose Wisely.pwp", "bench:level2/Pull Dont Push.pwp",
                    "cpptest:file_parsing.pwp", "bench:level2/Clean Sweep.pwp"]
    n_checked = n_pruned = n_solved = 0
    for key in keys:
        if key not in golden.meta:
            continue
        text = golden.text(key)
        oz = c_oracle.COraclePuzzle(text)
        cap = 1500 if chunk else 30000
        states, parent, action, depth, pruned, goal = host_iw(oz, width, max_states=cap + 1)
        if len(states) > cap:
            continue
        pz = PushWorldPuzzle(text=text)
        if lanes:
            pz._engine().set_option("step_kernel", "lane")
        bfs = BreadthFirstSearch(pz, max_states=cap + 8, novelty_width=width, chunk=chunk_arg)
        bfs.begin()
        while not bfs.exhausted:
            bfs.expand()
        assert bfs.total_states == len(states), key
        got = bfs.states()
        assert (got == np.array(states, dtype=np.int64).reshape(got.shape)).all(), key
        par, act = bfs.links()
        assert (par == np.array(parent)).all() and (act[1:] == np.array(action[1:])).all(), key
        assert (bfs.pruned() == np.array(pruned)).all(), key
        assert [c for _, c in bfs.layers] == np.bincount(np.array(depth)).tolist(), key
        assert bfs.goal_index == goal, key
        if goal >= 0:
            plan = bfs.plan(goal)
            assert len(plan) == depth[goal] and pz.is_valid_plan(plan)
            n_solved += 1
        n_pruned += int(np.sum(pruned))
        n_checked += 1
        bfs.close()
    assert n_checked >= 6 and n_pruned > 0 and n_solved >= 3


# ---------------------------------------------------------------------------------------------- pw_search_batch
def _host_bfs_verdict(oz, max_states):
    """(verdict, plan_len, states) of a layer-synchronous breadth-first search with pw_search_batch's rules: a goal state
    found in a layer decides (depth = plan length) even when the store fills up inside it; otherwise more than
    ``max_states`` states = unknown."""
    is_goal = oz.py.is_goal_state
    s0 = oz.initial_state
    if is_goal(s0):
        return 1, 0, 1
    seen = {s0}
    layer, depth = [s0], 0
    while layer:
        nxt, found = [], False
        for s in layer:
            for a in range(4):
                n = oz.get_next_state(s, a)
                if n == s or n in seen:
                    continue
                seen.add(n)
                nxt.append(n)
                found = found or is_goal(n)
        depth += 1
        if found:
            return 1, depth, len(seen)
        if len(seen) > max_states:
            return 2, -1, len(seen)
        layer = nxt
    return 0, -1, len(seen)


def test_search_batch_equals_a_host_breadth_first_search(golden):
    """One launch over a mixed set -- reference test puzzles, Level-0 puzzles, random puzzles, Level-1 puzzles of up to 8
    movables on up to 16 x 16 cells, a puzzle beyond the kernel's limits -- against a host breadth-first search over the
    oracle: verdict, shortest plan length and (for exhausted searches) the number of reachable states.  State caps of
    300 / 5 000 / 200 000 put puzzles on every level of the closed set (LDS, 2^16 and 2^20 slots) and on the `unknown` path."""
    from oracle import c_oracle
    from pushworld_amd import _capi
    from pushworld_amd.search import search_batch

    keys = [k for k in golden.keys if k.startswith(("pytest:", "cpptest:", "rand:"))][:60] + \
           [k for k in golden.keys if k.startswith("l0:")][::20] + \
           [k for k in golden.keys if k.startswith("bench:level1/") and golden.meta[k]["num_movables"] <= 8
            and golden.meta[k]["width"] <= 16 and golden.meta[k]["height"] <= 16][:12] + ["bench:level2/Clean Sweep.pwp"]
    texts = [golden.text(k) for k in keys]
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    eng = _capi.Engine(pset, None, 3, 1, _capi.OBS_U8)
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    seen_verdicts = set()
    for cap in (300, 5000, 200000):
        verdict, plan_len, n_states = search_batch(eng, None, max_states=cap)
        for i, k in enumerate(keys):
            m = golden.meta[k]
            if m["num_movables"] > 8 or m["width"] > 16 or m["height"] > 16:
                assert verdict[i] == 3 and plan_len[i] == -1, k
                continue
            if cap == 200000 and not k.startswith(("pytest:", "cpptest:", "l0:")):
                continue  # (the host search of a 200 000-state space in Python takes too long: the small caps cover these)
            want_v, want_len, want_states = _host_bfs_verdict(oracles[i], cap)
            assert (int(verdict[i]), int(plan_len[i])) == (want_v, want_len), (k, cap, int(n_states[i]), want_states)
            if want_v == 0:
                assert int(n_states[i]) == want_states, (k, cap)
            seen_verdicts.add(want_v)
    assert seen_verdicts == {0, 1, 2}
    # plans: A shortest plan of every solved puzzle, valid under the oracle (replayed from the initial state)
    v_p, l_p, _, plans = search_batch(eng, None, max_states=5000, plan_cap=64)
    v_all, p_all, _ = search_batch(eng, None, max_states=5000)
    # (with plans a goal state that found no room in the full store has no links to walk: such a search is `unknown`)
    lost = (v_all == 1) & (v_p == 2)
    assert ((v_p == v_all) | lost).all() and (l_p[~lost] == p_all[~lost]).all() and lost.sum() <= 3
    n_plans = 0
    for i, k in enumerate(keys):
        if v_p[i] != 1:
            assert plans[i] is None
            continue
        if l_p[i] > 64:
            assert plans[i] is None
            continue
        assert len(plans[i]) == l_p[i]
        s = oracles[i].initial_state
        for a in plans[i]:
            assert not oracles[i].py.is_goal_state(s), k  # (a shortest plan reaches the goal with its last action only)
            s = oracles[i].get_next_state(s, a)
        assert oracles[i].py.is_goal_state(s), k
        n_plans += 1
    assert n_plans >= 20
    # a subset by index, in another order
    sub = np.array([5, 0, 17, 3, 3], np.int32)
    v_all, p_all, _ = search_batch(eng, None, max_states=5000)
    v_sub, p_sub, _ = search_batch(eng, sub, max_states=5000)
    assert (v_sub == v_all[sub]).all() and (p_sub == p_all[sub]).all()


def test_batched_solvability_filter_equals_the_per_puzzle_search(torch_mod=None):
    """The filter of generate.py:262-297 over 2 000 device-generated Level-0 puzzles: ONE pw_search_batch launch gives the
    verdicts of the per-puzzle search (IW(2), then breadth-first search; a launch per pass and a readback per layer), and
    the shortest plan lengths of the per-puzzle breadth-first search on a sample."""
    import time

    from pushworld_amd import _capi, generate
    from pushworld_amd.search import BreadthFirstSearch, SetPuzzle, search_batch

    pset, grids, dims = generate.generate_level0_set(2000, device=0, random_seed=21)
    cap = 2_000_000  # (the library's default: no Level-0 state space of this recipe comes near it, so no verdict hinges on the cap)
    t0 = time.perf_counter()
    keep = generate.solvable_mask(pset, max_states=cap)
    t_batched = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = generate.solvable_mask(pset, max_states=cap, batched=False)
    t_single = time.perf_counter() - t0
    assert keep.tolist() == want.tolist() and 100 < keep.sum() < 2000
    print(f"solvability filter, 2 000 puzzles: batched {t_batched:.3f} s, per puzzle {t_single:.1f} s")
    eng = _capi.Engine(pset, None, 3, 1, _capi.OBS_U8)
    verdict, plan_len, n_states = search_batch(eng, None, max_states=cap)
    print("verdicts", np.bincount(verdict, minlength=4).tolist(), "states: median", int(np.median(n_states)), "max", int(n_states.max()),
          "sum", int(n_states.sum()))
    torch_sync = __import__("torch").cuda.synchronize
    t0 = time.perf_counter()
    for _ in range(3):
        search_batch(eng, None, max_states=cap)
    torch_sync()
    print(f"pw_search_batch alone (slabs allocated): {2000 * 3 / (time.perf_counter() - t0):.0f} puzzles/s")
    assert ((verdict == 1) <= keep).all() and ((verdict == 0) <= ~keep).all()  # (2 = unknown: decided by the per-puzzle search)
    for i in np.nonzero(verdict == 1)[0][:40]:
        bfs = BreadthFirstSearch(SetPuzzle(pset, int(i), eng), max_states=cap)
        bfs.begin()
        while bfs.goal_index < 0 and not bfs.exhausted:
            bfs.expand()
        assert bfs.goal_index >= 0 and len(bfs.plan(bfs.goal_index)) == int(plan_len[i]), int(i)
        bfs.close()
    for i in np.nonzero(verdict == 0)[0][:20]:
        bfs = BreadthFirstSearch(SetPuzzle(pset, int(i), eng), max_states=cap)
        bfs.begin()
        while not bfs.exhausted:
            bfs.expand()
        assert bfs.goal_index < 0 and bfs.total_states == int(n_states[i]), int(i)
        bfs.close()


def test_shortest_plan_is_one_launch_for_small_searches(golden):
    """``shortest_plan`` (pw_search_batch with n = 1) against the layer-synchronous BreadthFirstSearch on the small Level-1
    puzzles: same plan length (both breadth-first), a valid plan, and for `Choose Wisely` (59 states, 21 layers -- 0.9 ms
    through the per-layer launches) the whole search in a fraction of that."""
    import time

    import torch

    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch, shortest_plan

    for key in ("bench:level1/Choose Wisely.pwp", "bench:level1/2 Obstacle.pwp", "pytest:trivial_obstacle.pwp"):
        if key not in golden.meta:
            continue
        pz = PushWorldPuzzle(text=golden.text(key))
        plan, verdict = shortest_plan(pz)
        assert verdict == 1 and pz.is_valid_plan(plan), key
        bfs = BreadthFirstSearch(pz, max_states=1 << 20)
        bfs.begin()
        while bfs.goal_index < 0 and not bfs.exhausted:
            bfs.expand()
        assert len(bfs.plan(bfs.goal_index)) == len(plan), key
        bfs.close()
    pz = PushWorldPuzzle(text=golden.text("bench:level1/Choose Wisely.pwp"))
    shortest_plan(pz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        shortest_plan(pz)
    dt = (time.perf_counter() - t0) / 20
    print(f"Choose Wisely: shortest_plan {dt * 1e6:.0f} us per search")
    assert dt < 0.5e-3
