"""-m gpu: the latency path of the single-state API (``pw_next_state`` / ``pw_plan_states``) and the engine scratch sized at
creation (``PwEngineConfig::max_batch``).

``PushWorldPuzzle.get_next_state`` (puzzle.py:348-394), ``is_valid_plan`` (:413-424) and ``render_plan`` (:471-506) run one
launch each: the state travels in the kernel arguments, the result arrives in pinned host memory.  Checked against the
golden trajectories captured from the reference and against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_next_state_follows_every_golden_trajectory(golden):
    """Every step of every golden sequence (human plan, mid-plan walk, random walk) of 120 puzzles through
    ``get_next_state``; the whole sequence again through one ``pw_plan_states`` launch."""
    from pushworld_amd.puzzle import PushWorldPuzzle

    keys = [k for k in golden.keys if k.startswith(("bench:", "pytest:", "cpptest:"))][::3] + \
           [k for k in golden.keys if k.startswith("l0:")][::16]
    assert len(keys) >= 100
    n_steps = 0
    for k in keys:
        pz = PushWorldPuzzle(text=golden.text(k))
        for name, acts, start, pos, rew, term, goals in golden.sequences(k):
            st = pz.initial_state if start is None else tuple((int(x), int(y)) for x, y in start)
            n = pz.num_movables
            if len(acts) <= 120:
                s = st
                for t, a in enumerate(acts):
                    s = pz.get_next_state(s, int(a))
                    assert s == tuple((int(x), int(y)) for x, y in pos[t][:n]), (k, name, t)
                    n_steps += 1
            states, flags = pz._engine().plan_states(0, bytes(int(a) for a in acts), start=np.asarray(st, np.int8))
            assert states.shape == (len(acts) + 1, n, 2)
            assert (states[1:] == pos[:, :n].astype(np.int8)).all(), (k, name)
            assert (states[0] == np.asarray(st, np.int8)).all()
            assert (flags[1:] == term).all(), (k, name)
    assert n_steps > 1000


def test_next_state_info_and_errors(golden):
    import ctypes

    from pushworld_amd import _capi
    from pushworld_amd.puzzle import PushWorldPuzzle

    key = "bench:level1/2 Obstacle.pwp"
    pz = PushWorldPuzzle(text=golden.text(key))
    eng = pz._engine()
    n = pz.num_movables
    out = ctypes.create_string_buffer(64)
    info = (ctypes.c_int32 * 4)()
    plan = ["LRUD".index(c) for c in "URRRRUUUUULDDDDRULLLLLURRUURDDLDR"]
    s = pz.initial_state
    from oracle import pw_oracle
    oz = pw_oracle.OraclePuzzle(golden.text(key))
    for t, a in enumerate(plan):
        eng.next_state(0, bytes(v for p in s for v in p), a, out, info)
        want, moved = oz.get_next_state_moved(s, a) if hasattr(oz, "get_next_state_moved") else (oz.get_next_state(s, a), None)
        got = tuple((out.raw[2 * j], out.raw[2 * j + 1]) for j in range(n))
        assert got == tuple(want), t
        if moved is not None:
            assert info[0] == sum(1 << j for j in moved), t
        assert info[1] == oz.count_achieved_goals(s) and info[2] == oz.count_achieved_goals(want)
        assert info[3] == int(oz.is_goal_state(want))
        s = got
    assert info[3] == 1  # the human plan solves the puzzle
    with pytest.raises(ValueError):
        eng.next_state(0, bytes(2 * n), 4, out, None)
    with pytest.raises(ValueError):
        eng.next_state(7, bytes(2 * n), 0, out, None)
    with pytest.raises(ValueError):
        pz.get_next_state(pz.initial_state[:-1], 0)
    with pytest.raises(ValueError):
        pz.is_valid_plan([0, 5])
    assert _capi.lib.pw_plan_states(eng.handle, 0, None, None, -1, None, None, None) == _capi.PW_EINVAL


def test_get_next_states_is_the_batched_sibling(golden):
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle

    key = "bench:level2/Pull Dont Push.pwp"
    pz = PushWorldPuzzle(text=golden.text(key))
    oz = c_oracle.COraclePuzzle(golden.text(key))
    rng = np.random.default_rng(5)
    states = [pz.initial_state]
    for _ in range(400):  # a random walk supplies reachable states
        states.append(oz.get_next_state(states[-1], int(rng.integers(0, 4))))
    st = np.asarray(states, np.int8)
    acts = rng.integers(0, 4, size=len(states))
    got = pz.get_next_states(st, acts)
    assert got.shape == st.shape and got.dtype == np.int8
    for b in range(len(states)):
        assert tuple(map(tuple, got[b].tolist())) == oz.get_next_state(states[b], int(acts[b])), b
    assert pz.get_next_states(np.zeros((0, pz.num_movables, 2), np.int8), np.zeros((0,), np.int64)).shape == (0, pz.num_movables, 2)
    with pytest.raises(ValueError):
        pz.get_next_states(st, acts[:-1])
    big = np.repeat(st, 8, axis=0)  # beyond the first buffer size
    got = pz.get_next_states(big, np.repeat(acts, 8))
    assert (got[::8] == pz.get_next_states(st, acts)).all()


def test_render_plan_equals_frame_by_frame_renders(golden):
    from conftest import solution_plan
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle

    for key, ppc, bw in (("bench:level1/2 Obstacle.pwp", 20, 2), ("bench:level2/Pull Dont Push.pwp", 3, 1),
                         ("bench:level4/Four Pistons.pwp", 8, 2)):
        level, name = key.split(":", 1)[1].split("/")
        plan = solution_plan(level, name[:-4])
        pz = PushWorldPuzzle(text=golden.text(key))
        oz = c_oracle.COraclePuzzle(golden.text(key))
        frames = pz.render_plan(plan, border_width=bw, pixels_per_cell=ppc)
        assert len(frames) == len(plan) + 1
        s = oz.initial_state
        for t in range(len(plan) + 1):
            if t in (0, 1, len(plan) // 2, len(plan)):
                assert (frames[t] == oz.render(s, border_width=bw, pixels_per_cell=ppc)).all(), (key, t)
                assert (frames[t] == pz.render(s, border_width=bw, pixels_per_cell=ppc)).all(), (key, t)
            if t < len(plan):
                s = oz.get_next_state(s, plan[t])
        assert pz.is_valid_plan(plan) and not pz.is_valid_plan(plan[:-1]) and not pz.is_valid_plan(plan + [0])


def test_first_step_render_of_an_engine_is_capturable(golden):
    """``max_batch`` sizes the engine-owned scratch at creation: the very FIRST ``pw_step_render`` (and
    ``pw_step_render_delta``) call of an engine may sit inside a HIP graph capture (no hipMalloc / hipFree on the path)."""
    import torch

    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("bench:level1/")][::5]
    pool = [PushWorldPuzzle(text=golden.text(k)) for k in keys]
    B = 2048
    ids = np.arange(B) % len(pool)
    for incremental in (False, True):
        kw = dict(puzzle_ids=ids, max_steps=9, pixels_per_cell=3, border_width=1, observation="uint8", autoreset=True,
                  incremental=incremental, tune=False)
        eager, graphed = VecPushWorld(pool, B, **kw), VecPushWorld(pool, B, **kw)
        dev = eager.device
        acts = torch.randint(0, 4, (6, B), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        static = torch.zeros((B,), dtype=torch.uint8, device=dev)
        # nothing has run on `graphed`'s engine yet but its reset + first render (eager): capture its first step right away
        eager.reset()
        graphed.reset()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        static.copy_(acts[0])
        with torch.cuda.graph(g):
            graphed.step(static)
        for t in range(6):
            static.copy_(acts[t])
            g.replay()
            eager.step(acts[t])
            assert torch.equal(eager.pos, graphed.pos) and torch.equal(eager.obs, graphed.obs), (incremental, t)
    # ... and an engine created through the C ABI with max_batch allocates nothing in pw_step_render: its first call is captured
    from pushworld_amd import _capi

    pset = _capi.PuzzleSet([p._parsed for p in pool], 0)
    eng = _capi.Engine(pset, 9, 3, 1, _capi.OBS_U8, max_batch=B)
    st = eng.alloc_state(B)
    storage, view = eng.alloc_obs(B)
    pid = torch.as_tensor(ids, dtype=torch.int32).to(eng.device)
    eng.reset(pid, st["pos"], st["steps"], st["terminated"], st["truncated"])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.step_render(pid, static, st["pos"], st["steps"], st["reward"], st["dgoals"], st["terminated"], st["truncated"],
                        storage, _capi.STEP_AUTORESET)
    static.copy_(acts[0])
    g.replay()
    torch.cuda.synchronize()
    ref = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=9, pixels_per_cell=3, border_width=1, observation="uint8",
                       autoreset=True, tune=False)
    ref.reset()
    ref.step(acts[0])
    assert torch.equal(ref.pos, st["pos"]) and torch.equal(ref.obs, view)


def test_plans_longer_than_one_launch_continue_from_the_last_state(golden):
    """``PW_PLAN_MAX_ACTIONS`` (65 536) is a per-launch limit, not a limit of ``is_valid_plan`` / ``render_plan`` (the reference
    accepts plans of any length, puzzle.py:413-424, 471-506): 150 000 random actions through ``plan_states`` -- three launches,
    each continuing from the last state of the one before -- against the C oracle stepped action by action; the device copy
    of the states (what ``render_plan`` draws from) holds the same rows."""
    import torch

    from oracle import c_oracle
    from pushworld_amd import _capi
    from pushworld_amd.puzzle import PushWorldPuzzle

    text = golden.text("bench:level1/2 Obstacle.pwp")
    pz = PushWorldPuzzle(text=text)
    oz = c_oracle.COraclePuzzle(text)
    T = 150_000
    assert T > 2 * _capi.PLAN_MAX_ACTIONS
    acts = np.random.default_rng(21).integers(0, 4, size=T, dtype=np.uint8)
    eng = pz._engine()
    dev = torch.zeros((T + 1, eng.np, 2), dtype=torch.int8, device=eng.device)
    torch.cuda.current_stream(eng.device).synchronize()
    states, goals = eng.plan_states(0, bytes(acts), dev_states=dev)
    s = oz.initial_state
    n = oz.num_movables
    check_at = set(range(0, T + 1, 997)) | {_capi.PLAN_MAX_ACTIONS - 1, _capi.PLAN_MAX_ACTIONS, _capi.PLAN_MAX_ACTIONS + 1,
                                             2 * _capi.PLAN_MAX_ACTIONS, 2 * _capi.PLAN_MAX_ACTIONS + 1, T}
    goal_state = tuple(oz.py.goal_state)
    for t in range(T + 1):
        if t in check_at:
            assert tuple(map(tuple, states[t].tolist())) == tuple(s), t
            assert bool(goals[t]) == (tuple(s[1 : 1 + len(goal_state)]) == goal_state), t
        if t < T:
            s = oz.get_next_state(s, int(acts[t]))
    got_dev = dev.cpu().numpy()
    assert (got_dev[:, :n] == states).all() and (got_dev[:, n:] == 0).all()
    # and through the reference's own entry point: a long walk is not a valid plan, but it is answered, not refused
    assert pz.is_valid_plan([int(a) for a in acts[:70_000]]) in (True, False)
