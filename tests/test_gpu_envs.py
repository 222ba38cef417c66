"""-m gpu: the reference's own adapter tests (python3/test/test_puzzle.py, test_gym_env.py,
test_dm_env.py) re-expressed against pushworld_amd -- same puzzles, same expectations."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PUZZLES = os.path.join(ROOT, "tests", "puzzles", "ref_python")


def path(name):
    return os.path.join(PUZZLES, name + ".pwp")


# ----------------------------------------------------------------------- test_puzzle.py
def test_agent_movement():
    """test_puzzle.py:28-45 (free movement) + agent walls from a puzzle file instead of the
    injected collision-set entries of :47-66."""
    from pushworld_amd.puzzle import Actions, PushWorldPuzzle

    p = PushWorldPuzzle(path("agent_movement"))
    assert p.get_next_state(p.initial_state, Actions.LEFT)[0] == (1, 2)
    assert p.get_next_state(p.initial_state, Actions.RIGHT)[0] == (3, 2)
    assert p.get_next_state(p.initial_state, Actions.UP)[0] == (2, 1)
    assert p.get_next_state(p.initial_state, Actions.DOWN)[0] == (2, 3)
    boxed = PushWorldPuzzle(text=" .  AW  .\nAW   A AW\n .  AW  .\n")
    for a in range(4):
        assert boxed.get_next_state(boxed.initial_state, a)[0] == (2, 2)
    with pytest.raises(ValueError):
        p.get_next_state(p.initial_state, 4)


def test_pushing_and_transitive_pushing():
    """test_puzzle.py:69-98."""
    from pushworld_amd.puzzle import Actions, PushWorldPuzzle

    p = PushWorldPuzzle(path("pushing"))
    assert p.get_next_state(p.initial_state, Actions.DOWN) == ((1, 2), (2, 1))
    s = p.get_next_state(p.initial_state, Actions.RIGHT)
    assert s == ((2, 1), (3, 1))
    s = p.get_next_state(s, Actions.RIGHT)
    assert s == ((3, 1), (4, 1))
    assert p.get_next_state(s, Actions.RIGHT) == ((3, 1), (4, 1))  # transitive stopping

    p = PushWorldPuzzle(path("transitive_pushing"))
    assert p.get_next_state(p.initial_state, Actions.DOWN) == ((1, 2), (5, 1), (3, 1))
    s = p.get_next_state(p.initial_state, Actions.RIGHT)
    assert s == ((2, 1), (5, 1), (3, 1))
    s = p.get_next_state(s, Actions.RIGHT)
    assert s == ((3, 1), (5, 1), (4, 1))
    s = p.get_next_state(s, Actions.RIGHT)
    assert s == ((4, 1), (6, 1), (5, 1))
    s = p.get_next_state(s, Actions.RIGHT)
    assert s == ((4, 1), (6, 1), (5, 1))
    assert p.get_next_state(s, Actions.DOWN) == ((4, 2), (6, 1), (5, 1))


def test_goal_states():
    """test_puzzle.py:101-123."""
    from pushworld_amd.puzzle import PushWorldPuzzle

    p = PushWorldPuzzle(path("is_goal_state"))
    for state, goal, count in [
        (((5, 1), (3, 6), (2, 5)), True, 2),
        (((2, 8), (3, 6), (2, 5)), True, 2),
        (((1, 1), (3, 3), (2, 5)), False, 1),
        (((1, 1), (3, 6), (2, 2)), False, 1),
        (((1, 1), (3, 4), (1, 5)), False, 0),
    ]:
        assert p.is_goal_state(state) == goal
        assert p.count_achieved_goals(state) == count


def test_trivial_trajectory_and_plan_validation():
    """test_puzzle.py:126-194 (trajectory) and cpp test_pushworld_puzzle.cc:391-393 / puzzle.py:413-424."""
    from pushworld_amd.puzzle import Actions as A
    from pushworld_amd.puzzle import PushWorldPuzzle

    p = PushWorldPuzzle(path("trivial"))
    assert p.goal_state == ((3, 1),) and p.initial_state == ((1, 2), (2, 2))
    s = p.initial_state
    expect = [
        (A.LEFT, ((1, 2), (2, 2)), False), (A.UP, ((1, 2), (2, 2)), False), (A.DOWN, ((1, 2), (2, 2)), False),
        (A.RIGHT, ((2, 2), (3, 2)), False), (A.RIGHT, ((2, 2), (3, 2)), False), (A.DOWN, ((2, 3), (3, 2)), False),
        (A.DOWN, ((2, 3), (3, 2)), False), (A.RIGHT, ((3, 3), (3, 2)), False), (A.RIGHT, ((3, 3), (3, 2)), False),
        (A.UP, ((3, 2), (3, 1)), True), (A.UP, ((3, 2), (3, 1)), True),
    ]
    for a, want, goal in expect:
        s = p.get_next_state(s, a)
        assert s == want and p.is_goal_state(s) == goal
    assert p.is_valid_plan([A.RIGHT, A.DOWN, A.RIGHT, A.UP])
    assert not p.is_valid_plan([A.RIGHT, A.DOWN, A.LEFT, A.UP])
    # the Python reference rejects plans that reach the goal before their last action
    assert not p.is_valid_plan([A.RIGHT, A.DOWN, A.RIGHT, A.UP, A.UP])
    assert not p.is_valid_plan([])


def test_file_parsing_properties():
    """test_puzzle.py:197-211 + property semantics (trap T2: agent_wall_positions = AW u W)."""
    from pushworld_amd.puzzle import Colors, PushWorldPuzzle

    p = PushWorldPuzzle(path("file_parsing"))
    assert p.dimensions == (12, 18)
    assert p.goal_state == ((6, 5), (3, 4))
    assert p.initial_state == ((1, 12), (6, 14), (1, 3), (4, 1), (2, 7), (3, 8))
    assert p.num_movables == 6 and len(p.movable_objects) == 6
    assert p.movable_objects[0].fill_color == Colors.AGENT
    assert p.movable_objects[1].fill_color == Colors.GOAL_OBJECT and p.movable_objects[2].fill_color == Colors.GOAL_OBJECT
    assert p.movable_objects[3].fill_color == Colors.MOVABLE
    assert p.wall_positions <= p.agent_wall_positions
    assert {(8, 8), (8, 9)} <= p.agent_wall_positions and (8, 8) not in p.wall_positions


def test_rendering_hashes(golden):
    """test_puzzle.py:249-271: hash(tuple(image.flat)) of the 5 frames along R, D, R, U."""
    from pushworld_amd.puzzle import Actions, PushWorldPuzzle

    p = PushWorldPuzzle(path("trivial"))
    initial = p.render(p.initial_state)
    assert initial.shape == (100, 100, 3) and initial.dtype == np.uint8
    frames = p.render_plan([Actions.RIGHT, Actions.DOWN, Actions.RIGHT, Actions.UP])
    assert (initial == frames[0]).all()
    assert [hash(tuple(int(v) for v in f.flat)) for f in frames] == golden.ref_render_hashes
    with pytest.raises(ValueError):
        p.render(p.initial_state, border_width=0)
    with pytest.raises(ValueError):
        p.render(p.initial_state, border_width=2, pixels_per_cell=4)


def test_dataset_solutions_are_valid_plans():
    """test_dataset.py:24-61: every benchmark puzzle has a human plan that is_valid_plan accepts
    (all 223 plans, 18 143 actions, replayed on the GPU)."""
    from conftest import DATA, solution_plan
    from pushworld_amd.puzzle import PushWorldPuzzle

    n = 0
    for level in ("level1", "level2", "level3", "level4"):
        d = os.path.join(DATA, "puzzles", level)
        for f in sorted(os.listdir(d)):
            if f.endswith(".pwp"):
                p = PushWorldPuzzle(os.path.join(d, f))
                assert p.is_valid_plan(solution_plan(level, f[:-4])), f
                n += 1
    assert n == 223


# ---------------------------------------------------------------------- test_gym_env.py
@pytest.fixture(params=["gym", "dm"])
def flavour(request):
    return request.param


def _make(flavour, *args, **kw):
    if flavour == "gym":
        from pushworld_amd.gym_env import PushWorldEnv
    else:
        from pushworld_amd.dm_env import PushWorldEnv
    return PushWorldEnv(*args, **kw)


def _step(flavour, env, action):
    """Normalises both adapters to (obs, reward, terminated_or_last, truncated_or_None, state)."""
    if flavour == "gym":
        obs, reward, terminated, truncated, info = env.step(action)
        return obs, reward, terminated, truncated, info["puzzle_state"]
    ts = env.step(action)
    return ts.observation, ts.reward, ts.last(), None, env.current_state


def _reset(flavour, env, **kw):
    if flavour == "gym":
        obs, info = env.reset(**kw)
        return obs, info["puzzle_state"]
    ts = env.reset(**kw)
    assert ts.first() and ts.reward is None and ts.discount is None
    return ts.observation, env.current_state


def test_observations_and_renderings(flavour):
    """test_gym_env.py:28-44 / test_dm_env.py:28-47."""
    from pushworld_amd.puzzle import Actions

    env = _make(flavour, path("trivial"))
    obs, state = _reset(flavour, env)
    assert obs.dtype == np.float32 and obs.shape == (100, 100, 3) and obs.min() >= 0 and obs.max() <= 1
    if flavour == "gym":
        assert obs in env.observation_space
    else:
        env.observation_spec().validate(obs)
    image = env.current_puzzle.render(state)
    assert (image == obs * 255).all()
    ren = env.render()
    if flavour == "gym":
        assert ren.dtype == np.uint8 and (image == ren).all()
    else:
        assert ren.dtype == np.float32 and (ren == obs).all()
    obs, _, _, _, state = _step(flavour, env, Actions.RIGHT)
    image = env.current_puzzle.render(state)
    assert (image == obs * 255).all()
    assert (obs == (image.astype(np.float32) / 255)).all()


def test_standard_padding(flavour):
    """test_gym_env.py:47-64: padding changes the shape, not the non-zero content."""
    env = _make(flavour, path("trivial"), standard_padding=False)
    o1, _ = _reset(flavour, env)
    env2 = _make(flavour, path("trivial"), standard_padding=True)
    o2, _ = _reset(flavour, env2)
    assert o1.shape == (100, 100, 3) and o2.shape == (54 * 20, 47 * 20, 3)
    assert np.count_nonzero(o1.sum(axis=0)) == np.count_nonzero(o2.sum(axis=0))
    top, left = (o2.shape[0] - 100) // 2, (o2.shape[1] - 100) // 2
    assert (o2[top : top + 100, left : left + 100] == o1).all()


def test_reward(flavour):
    """test_gym_env.py:67-84: exact python floats -0.01, 0.99, -1.01, 10.0."""
    from pushworld_amd.puzzle import Actions as A

    env = _make(flavour, path("multiple_goals"))
    _reset(flavour, env)
    assert _step(flavour, env, A.RIGHT)[1] == -0.01
    assert _step(flavour, env, A.RIGHT)[1] == 1 - 0.01
    assert _step(flavour, env, A.RIGHT)[1] == -1 - 0.01
    _reset(flavour, env)
    for a in (A.RIGHT, A.RIGHT, A.LEFT, A.LEFT, A.LEFT):
        _step(flavour, env, a)
    r = _step(flavour, env, A.LEFT)[1]
    assert r == 10 and isinstance(r, float)


@pytest.mark.parametrize("standard_padding", [True, False])
def test_all_goals_achieved(flavour, standard_padding):
    """test_gym_env.py:87-103 / test_dm_env.py:87-101."""
    from pushworld_amd.puzzle import Actions as A

    env = _make(flavour, path("trivial"), standard_padding=standard_padding)
    _reset(flavour, env)
    for a in (A.RIGHT, A.DOWN, A.RIGHT):
        _step(flavour, env, a)
    obs, reward, done, truncated, _ = _step(flavour, env, A.UP)
    assert reward == 10 and done
    if flavour == "gym":
        assert truncated is False and obs in env.observation_space


def test_truncation_and_termination(flavour):
    """test_gym_env.py:106-152 / test_dm_env.py:104-153."""
    from pushworld_amd.puzzle import Actions as A

    env = _make(flavour, path("transitive_pushing"), max_steps=3)
    _reset(flavour, env)
    if flavour == "gym":
        assert env.step(A.LEFT)[3] is False and env.step(A.LEFT)[3] is False and env.step(A.LEFT)[3] is True
    else:
        assert env.step(A.LEFT).mid() and env.step(A.LEFT).mid()
        ts = env.step(A.LEFT)
        assert ts.last() and ts.reward != 10 and ts.discount == 0.0
    env = _make(flavour, path("transitive_pushing"))
    _reset(flavour, env)
    _step(flavour, env, A.RIGHT)
    _step(flavour, env, A.RIGHT)
    _, reward, done, truncated, _ = _step(flavour, env, A.RIGHT)
    assert reward == 10 and done and truncated in (False, None)
    env = _make(flavour, path("multiple_goals"))
    for action in (A.LEFT, A.RIGHT):
        _reset(flavour, env)
        _step(flavour, env, action)
        _, reward, done, _, _ = _step(flavour, env, action)
        assert reward > 0 and not done
    # no auto-reset: stepping after termination keeps simulating (trap T9)
    env = _make(flavour, path("trivial"))
    _reset(flavour, env)
    for a in (A.RIGHT, A.DOWN, A.RIGHT, A.UP):
        out = _step(flavour, env, a)
    assert out[2]
    out = _step(flavour, env, A.DOWN)
    assert out[4] == ((3, 3), (3, 1)) and out[1] == 10.0  # goal object still on its goal


def test_reset_and_errors(flavour):
    """test_gym_env.py:155-169 + error conventions (gym_env.py:71-76,195-199)."""
    from pushworld_amd.puzzle import Actions as A

    env = _make(flavour, path("trivial"))
    with pytest.raises(RuntimeError):
        env.step(A.LEFT)
    o1, _ = _reset(flavour, env)
    o2 = _step(flavour, env, A.RIGHT)[0]
    o3, _ = _reset(flavour, env)
    assert not (o1 == o2).all() and (o1 == o3).all()
    for bad in (4, -1, 1.5):
        with pytest.raises(ValueError):
            env.step(bad)
    with pytest.raises(ValueError):
        _make(flavour, path("trivial"), border_width=0)
    with pytest.raises(ValueError):
        _make(flavour, path("trivial"), pixels_per_cell=2)
    with pytest.raises(ValueError):
        _make(flavour, path("trivial"), pixels_per_cell=4, border_width=2)
    with pytest.raises(ValueError):
        _make(flavour, os.path.join(ROOT, "include"))
    pool = _make(flavour, PUZZLES, pixels_per_cell=3, border_width=1)
    firsts = set(tuple(_reset(flavour, pool)[0].flat) for _ in range(40))
    assert len(firsts) > 1
    # seeded resets pick the same puzzle as random.Random(seed).choice over the os.walk pool
    import random

    from pushworld_amd.config import PUZZLE_EXTENSION
    from pushworld_amd.utils.filesystem import iter_files_with_extension

    files = list(iter_files_with_extension(PUZZLES, PUZZLE_EXTENSION))
    for seed in (0, 7, 123):
        _reset(flavour, pool, seed=seed)
        assert pool.current_puzzle.file_path == random.Random(seed).choice(files)


# ----------------------------------------------------------------------- the adapters' step as one launch (round 6)
@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("level1", [False, True])
def test_single_env_step_one_launch_against_the_oracle(fused, level1, tmp_path):
    """gym_env.py:188-226 through pw_step_render_delta on a batch of one: with PW_OPT_STEP_ONE_FUSED (default) the step and the
    redraw of the rows it changed are ONE launch (workgroup 0 steps, seven more draw), without it two launches / a graph replay.
    EVERY step's observation, state, reward and flags against the oracle -- an incremental redraw that misses a row shows up the
    step it happens -- on the C1 puzzle and on a Level-1 puzzle with pushes and transitive pushes, resets included."""
    import glob

    from oracle import pw_oracle
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.config import BENCHMARK_PUZZLES_PATH
    from pushworld_amd.gym_env import PushWorldEnv

    if level1:
        src = sorted(glob.glob(os.path.join(BENCHMARK_PUZZLES_PATH, "level1", "*.pwp")))[7]
        text = open(src).read()
    else:
        text = next(iter(bd.level0_texts(("base",), "train", 1).values()))
    f = tmp_path / "one.pwp"
    f.write_text(text)
    env = PushWorldEnv(str(f), max_steps=40)
    assert env._engine.get_option("step_one_fused") == 1
    env._engine.set_option("step_one_fused", fused)
    oz = pw_oracle.OraclePuzzle(text)
    oenv = pw_oracle.OracleEnv(oz, max_steps=40)
    rng = np.random.default_rng(5 + fused)
    obs, info = env.reset(seed=0)
    ostate = oenv.reset()
    assert info["puzzle_state"] == ostate and (obs == oz.observation(ostate, oz.height, oz.width)).all()
    held = obs  # (a returned observation is the caller's: later steps must not change it)
    held_copy = obs.copy()
    for t in range(260 if level1 else 400):
        a = int(rng.integers(0, 4)) if t % 7 else 1  # (runs against walls too: steps that change nothing)
        obs, r, term, trunc, info = env.step(a)
        ostate, orew, oterm, otrunc = oenv.step(a)
        assert info["puzzle_state"] == ostate and r == orew and term == oterm and trunc == otrunc, t
        assert (obs == oz.observation(ostate, oz.height, oz.width)).all(), t
        if term or trunc:
            obs, info = env.reset()
            ostate = oenv.reset()
            assert (obs == oz.observation(ostate, oz.height, oz.width)).all(), t
    assert (held == held_copy).all()
    if fused:
        assert env._graphs is False  # (one launch per step: no graph is captured -- its hand-over word carries a launch number)
