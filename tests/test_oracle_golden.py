"""The oracle (oracle/pw_oracle.py and its C twin oracle/pw_oracle.c) pinned against fixtures
captured from the reference itself (tests/golden/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

from oracle import c_oracle, pw_oracle


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def digest_points(points):
    return sha(np.array(sorted(points), dtype=np.int32).reshape(-1, 2))


def _subset(golden, l0_stride=4):
    keys = []
    for i, k in enumerate(golden.keys):
        if k.startswith("l0:") and i % l0_stride:
            continue
        keys.append(k)
    return keys


def test_python_oracle_parse_and_tables(golden):
    """Parse products and the full collision tables (sizes + SHA-256 of the sorted contents)."""
    for k in _subset(golden, 8):
        m = golden.meta[k]
        o = pw_oracle.OraclePuzzle(golden.text(k))
        assert (o.width, o.height, o.num_movables) == (m["width"], m["height"], m["num_movables"]), k
        assert [list(p) for p in o.initial_state] == m["initial_state"], k
        assert [list(p) for p in o.goal_state] == m["goal_state"], k
        assert [sorted(map(list, s)) for s in o.shapes] == m["object_cells"], k
        assert [sorted(map(list, s)) for s in o.goal_shapes] == m["goal_cells"], k
        assert digest_points(o.wall_cells) == m["walls_sha"]
        assert digest_points(o.agent_wall_cells) == m["agent_walls_prop_sha"]
        assert o.has_agent_walls == m["has_agent_walls"]
        n = o.num_movables
        h = hashlib.sha256()
        for a in range(4):
            for i in range(n):
                assert len(o.static[a][i]) == m["static_sizes"][a][i], (k, a, i)
                h.update(np.array(sorted(o.static[a][i]), np.int32).tobytes())
                for j in range(n):
                    assert len(o.dynamic[a][i][j]) == m["dynamic_sizes"][a][i][j], (k, a, i, j)
                    h.update(np.array(sorted(o.dynamic[a][i][j]), np.int32).tobytes())
        assert h.hexdigest() == m["tables_sha"], k


def test_c_oracle_table_sizes(golden):
    for k in _subset(golden, 8):
        m = golden.meta[k]
        c = c_oracle.COraclePuzzle(golden.text(k))
        s, d = c.table_sizes()
        assert s.tolist() == m["static_sizes"], k
        assert d.tolist() == m["dynamic_sizes"], k


@pytest.mark.parametrize("impl", ["python", "c"])
def test_oracle_trajectories(golden, impl):
    """Positions, float64 reward bits, terminated and goal counts of every golden sequence."""
    for k in _subset(golden, 4 if impl == "c" else 8):
        text = golden.text(k)
        o = pw_oracle.OraclePuzzle(text) if impl == "python" else c_oracle.COraclePuzzle(text)
        po = o if impl == "python" else o.py
        for name, acts, start, pos, rew, term, goals in golden.sequences(k):
            state = po.initial_state if start is None else tuple(map(tuple, start.tolist()))
            for t, a in enumerate(acts):
                prev = state
                state = o.get_next_state(state, int(a))
                assert [list(p) for p in state] == pos[t].tolist(), (k, name, t)
                terminated = po.is_goal_state(state)
                reward = 10.0 if terminated else po.count_achieved_goals(state) - po.count_achieved_goals(prev) - 0.01
                assert np.float64(reward).view(np.uint64) == rew[t].view(np.uint64), (k, name, t)
                assert terminated == bool(term[t])
                assert po.count_achieved_goals(state) == goals[t]
            if impl == "c":  # the C env step (reward formed in C double arithmetic)
                state = po.initial_state if start is None else tuple(map(tuple, start.tolist()))
                for t, a in enumerate(acts[:64]):
                    state, r, te = o.env_step(state, int(a))
                    assert np.float64(r).view(np.uint64) == rew[t].view(np.uint64)
                    assert te == bool(term[t])


@pytest.mark.parametrize("impl", ["python", "c"])
def test_oracle_overlapping_states(golden, impl):
    keys = [k for k in golden.keys if f"{k}|in" in golden.states]
    for k in keys[:: (2 if impl == "c" else 5)]:
        text = golden.text(k)
        o = pw_oracle.OraclePuzzle(text) if impl == "python" else c_oracle.COraclePuzzle(text)
        sin, sout = golden.states[f"{k}|in"], golden.states[f"{k}|out"]
        for s in range(sin.shape[0]):
            st = tuple(map(tuple, sin[s].tolist()))
            for a in range(4):
                assert [list(p) for p in o.get_next_state(st, a)] == sout[s, a].tolist(), (k, s, a)


@pytest.mark.parametrize("impl", ["python", "c"])
def test_oracle_render_digests(golden, impl):
    """uint8 images and padded float32 observations against the reference's SHA-256 digests."""
    pads = {"own": None, "l1": (51, 42), "std": (54, 47)}
    n = 0
    keys = [k for k in golden.keys if golden.meta[k]["renders"]]
    stride = 1 if impl == "c" else 6
    for k in keys[::stride]:
        m = golden.meta[k]
        text = golden.text(k)
        o = pw_oracle.OraclePuzzle(text, build_tables=False) if impl == "python" else c_oracle.COraclePuzzle(text)
        for ent in m["renders"]:
            ppc, bw = ent["ppc"], ent["bw"]
            if impl == "python" and ppc == 20 and ent["seq"] != "init":
                continue
            st = [tuple(p) for p in ent["state"]]
            assert sha(o.render(st, bw, ppc)) == ent["u8"], (k, ent["seq"], ppc)
            for pname, pad in pads.items():
                if f"f32_{pname}" not in ent or (ppc == 20 and pname != "own" and ent["seq"] != "init"):
                    continue
                mh, mw = (m["height"], m["width"]) if pad is None else pad
                assert sha(o.observation(st, mh, mw, ppc, bw)) == ent[f"f32_{pname}"], (k, pname, ppc)
            n += 1
    assert n > 50


def test_reference_render_hash_test(golden):
    """python3/test/test_puzzle.py:249-271 re-expressed: hash(tuple(image.flat)) of the 5
    frames of trivial.pwp along R, D, R, U."""
    o = pw_oracle.OraclePuzzle(golden.text("pytest:trivial.pwp"))
    frames = [o.render(o.initial_state)]
    s = o.initial_state
    for a in (1, 3, 1, 2):
        s = o.get_next_state(s, a)
        frames.append(o.render(s))
    assert frames[0].shape == (100, 100, 3)
    assert [hash(tuple(int(v) for v in f.flat)) for f in frames] == golden.ref_render_hashes
    assert (np.stack(frames) == golden.images["pytest:trivial.pwp|render_plan_RDRU"]).all()


def test_cpp_order_known_answers(golden):
    """cpp/test/test_pushworld_puzzle.cc:461-514 (object order + table sizes of
    file_parsing.pwp) and :260-394 (trivial.pwp) re-expressed for order="cpp"."""
    o = pw_oracle.OraclePuzzle(golden.text("cpptest:file_parsing.pwp"), order="cpp")
    assert o.goal_state == ((3, 4), (6, 5))
    assert o.initial_state == ((1, 12), (1, 3), (6, 14), (4, 1), (2, 7), (3, 8))
    L, R, U, D = 0, 1, 2, 3
    assert [len(o.static[L][i]) for i in range(6)] == [16, 16, 15, 15, 14, 16]
    assert [len(o.static[R][i]) for i in range(6)] == [16, 16, 15, 15, 14, 16]
    assert [len(o.static[U][i]) for i in range(6)] == [9, 10, 9, 9, 8, 10]
    assert len(o.dynamic[D][0][4]) == 5 and len(o.dynamic[D][0][3]) == 4
    for a in (D, L, R, U):
        assert len(o.dynamic[a][1][2]) == 2 and len(o.dynamic[a][1][4]) == 4

    t = pw_oracle.OraclePuzzle(golden.text("cpptest:trivial.pwp"), order="cpp")
    assert t.goal_state == ((3, 1),) and t.initial_state == ((1, 2), (2, 2))
    assert t.static[L][0] == {(2, 1), (1, 2), (2, 3)}
    assert t.static[U][0] == {(1, 2), (2, 1), (3, 1)}
    assert t.static[R][0] == {(3, 1), (3, 2), (3, 3)}
    assert t.static[D][0] == {(1, 2), (2, 3), (3, 3)}
    assert t.dynamic[L][0][1] == {(1, 0)} and t.dynamic[R][0][1] == {(-1, 0)}
    assert t.dynamic[U][0][1] == {(0, 1)} and t.dynamic[D][0][1] == {(0, -1)}
    # moved_object_indices: agent first, ascending; empty when nothing moves
    s, moved = t.get_next_state_moved(t.initial_state, L)
    assert s == t.initial_state and moved == []
    s, moved = t.get_next_state_moved(t.initial_state, R)
    assert s == ((2, 2), (3, 2)) and moved == [0, 1]
    assert t.is_valid_plan([R, D, R, U]) and t.is_valid_plan([R, D, R, D, R, U], reject_early_goal=False)
    assert not t.is_valid_plan([R, D, L, U])

    ov = pw_oracle.OraclePuzzle(golden.text("cpptest:trivial_overlap.pwp"), order="cpp")
    assert ov.goal_state == ((2, 1),)
    assert ov.initial_state == ((2, 1), (1, 1), (2, 2))
    assert [len(ov.static[a][0]) for a in range(4)] == [2, 2, 2, 2]
    assert [len(ov.dynamic[a][0][1]) for a in (L, R, U, D)] == [2, 2, 1, 1]
    assert [len(ov.dynamic[a][2][1]) for a in (L, R, U, D)] == [2, 2, 1, 1]
    # the remaining literal expectations of cc:400-458: members of the agent's static sets, sizes for object 2
    assert ov.static[L][0] == {(1, 1), (1, 2)} and ov.static[U][0] == {(1, 1), (2, 1)}
    assert ov.static[R][0] == {(2, 1), (1, 2)} and ov.static[D][0] == {(2, 1), (1, 2)}
    assert [len(ov.dynamic[a][0][2]) for a in (L, R, U, D)] == [1, 1, 1, 1]


@pytest.mark.parametrize("impl", ["python", "c"])
def test_cpp_hand_built_collision_cases(impl):
    """cpp/test/test_pushworld_puzzle.cc:84-257 (agent movement with hand-set agent walls, pushing, transitive
    pushing with moved_object_indices, satisfiesGoal), re-expressed as small puzzles: tests/cpp_cases.py."""
    import cpp_cases

    def make(text):
        return c_oracle.COraclePuzzle(text, order="cpp") if impl == "c" else pw_oracle.OraclePuzzle(text, order="cpp")

    n = 0
    for name, text, steps in cpp_cases.movement_cases():
        o = make(text)
        assert tuple(o.initial_state) == steps[0][0], name
        for state, action, want, moved in steps:
            got, got_moved = o.get_next_state_moved(state, action)
            assert got == want, (name, state, action)
            if moved is not None:
                assert list(got_moved) == moved, (name, state, action)   # agent first, ascending, empty if blocked
            n += 1
    assert n == 22
    for name, text, checks in cpp_cases.goal_cases():
        o = pw_oracle.OraclePuzzle(text, order="cpp")
        assert o.initial_state == ((1, 1), (2, 2), (3, 3)), name
        for state, want in checks:
            assert o.is_goal_state(state) == want, (name, state)


def test_python_and_cpp_orders_agree_up_to_permutation(golden):
    """The two reference engines are the same dynamics under an object permutation (T1)."""
    rng = np.random.default_rng(7)
    for k in [k for k in golden.keys if k.startswith("bench:")][::9]:
        text = golden.text(k)
        a_ = pw_oracle.OraclePuzzle(text, order="python")
        b_ = pw_oracle.OraclePuzzle(text, order="cpp")
        perm = [b_.names.index(n) for n in a_.names]
        sa, sb = a_.initial_state, b_.initial_state
        for act in rng.integers(0, 4, size=150):
            sa = a_.get_next_state(sa, int(act))
            sb = b_.get_next_state(sb, int(act))
            assert tuple(sb[p] for p in perm) == sa, k
            assert a_.is_goal_state(sa) == b_.is_goal_state(sb)


def test_novelty_oracle_known_answers():
    """cpp/test/heuristics/test_novelty_heuristic.cc:86-106 transcribed as data: the sequence of
    (state, moved indices) -> novelty the reference asserts for NoveltyHeuristic(4)."""
    nov = pw_oracle.OracleNovelty(4)
    seq = [((1, 2, 3, 4), (0, 1, 2, 3), 1), ((2, 3, 4, 5), (0, 1, 2, 3), 1), ((1, 3, 4, 5), (0,), 2),
           ((2, 3, 3, 5), (2,), 2), ((1, 3, 3, 5), (0, 2), 3), ((1, 3, 3, 4), (3,), 2), ((1, 3, 5, 4), (2,), 1),
           ((1, 3, 5, 4), (), 3)]
    for state, moved, want in seq:
        assert nov.estimate(state, moved) == want, (state, moved)


def _cpp_order_fixture():
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_cpp_order.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("impl", ["py", "c"])
def test_cpp_order_with_many_ids_against_the_reference(impl):
    """tests/golden/golden_cpp_order.json (make_cpp_order_golden.py): 24 random puzzles with 10-13 movables whose ids run past 10,
    stepped by the PYTHON reference and rearranged into the C++ object order of pushworld_puzzle.cc:262-321 (goal ids ascending
    by string: g1 < g10 < g2; then the other movables the same way).  Both oracles in order="cpp": object names, initial and goal
    state, 300 steps of a random walk, satisfiesGoal of every state."""
    fx = _cpp_order_fixture()
    assert len(fx) >= 20
    for key, ent in fx.items():
        text = ent["text"]
        o = pw_oracle.OraclePuzzle(text, order="cpp", build_tables=impl == "py")
        assert o.names == ent["cpp_names"], key
        assert any(int(nm[1:]) >= 10 for nm in o.names[1:]) and ent["cpp_names"] != ent["python_names"]
        assert [list(p) for p in o.initial_state] == ent["states_cpp"][0], key
        assert [list(p) for p in o.goal_state] == ent["goal_state_cpp"], key
        stepper = c_oracle.COraclePuzzle(text, order="cpp") if impl == "c" else o
        s = tuple(tuple(p) for p in ent["states_cpp"][0])
        for t, a in enumerate(ent["actions"]):
            s = tuple(tuple(p) for p in stepper.get_next_state(s, a))
            assert [list(p) for p in s] == ent["states_cpp"][t + 1], (key, t)
            assert o.is_goal_state(s) == ent["goal_flags"][t + 1], (key, t)
        for smp in ent["expand"]:
            st = tuple(tuple(p) for p in ent["states_cpp"][smp["t"]])
            for b in range(4):
                if impl == "c":
                    nxt, moved = stepper.get_next_state_moved(st, b)
                    assert sum(1 << k for k in moved) == smp["moved"][b], (key, smp["t"], b)
                else:
                    nxt = stepper.get_next_state(st, b)
                assert [list(p) for p in nxt] == smp["succ"][b], (key, smp["t"], b)
