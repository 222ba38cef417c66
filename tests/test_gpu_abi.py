"""-m gpu: the parts of the C ABI that are not dynamics -- engine options (former environment variables),
pw_validate_state, the sticky bad-action counter, max_steps = 0, the render profiler and the rule that no entry
point changes the caller's current HIP device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _level1_vec(B=512, **kw):
    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    args = dict(puzzle_ids=ids, max_steps=50, pixels_per_cell=3, border_width=1, observation="uint8", device=0,
                autoreset=True)
    args.update(kw)
    return VecPushWorld([PushWorldPuzzle(p) for p in paths], B, **args)


def test_engine_options_round_trip_and_select_kernels(torch_mod):
    from pushworld_amd import _capi

    vec = _level1_vec()
    eng = vec.engine
    for name in ("step_kernel", "fused_step_render", "render_kernel", "page_slice_envs", "search_chunk", "profile_render"):
        assert eng.get_option(name) == 0
    # launch configuration of the page kernel: the robust default until the tuner has run
    assert (eng.get_option("page_order"), eng.get_option("page_run_log2"), eng.get_option("page_lds_pad_kb")) == (2, 6, 7)
    assert eng.render_kernel == "pw_render_page_kernel"
    eng.set_option("render_kernel", "lds")
    assert eng.get_option("render_kernel") == 1 and eng.render_kernel == "pw_render_u8_ppc3_kernel"
    eng.set_option("render_kernel", "auto")
    eng.set_option("step_kernel", "lane")
    assert eng.get_option("step_kernel") == 2
    eng.set_option("page_slice_envs", 100)
    assert eng.get_option("page_slice_envs") == 100
    for bad in (("step_kernel", 3), ("render_kernel", 2), ("page_slice_envs", -1), ("page_order", 3), ("page_lds_pad_kb", 49),
                (99, 0)):
        with pytest.raises(ValueError):
            eng.set_option(*bad)
    # options never change results: same walk with every combination
    torch = torch_mod
    ref = _level1_vec()
    alt = _level1_vec(engine_options={"step_kernel": "wave", "render_kernel": "lds"})
    o0, o1 = ref.reset(), alt.reset()
    assert torch.equal(o0, o1)
    g = torch.Generator(device=ref.device).manual_seed(3)
    for t in range(12):
        a = torch.randint(0, 4, (ref.num_envs,), generator=g, device=ref.device, dtype=torch.uint8)
        r0, r1 = ref.step(a), alt.step(a)
        for x, y in zip(r0, r1):
            assert torch.equal(x, y), t
        assert torch.equal(ref.pos, alt.pos)


def test_render_profiler_times_every_launch(torch_mod):
    torch = torch_mod
    vec = _level1_vec(B=2048)
    vec.reset()
    eng = vec.engine
    assert eng.profile_read() == []
    eng.profile_render(5)
    a = torch.zeros((vec.num_envs,), dtype=torch.uint8, device=vec.device)
    for _ in range(7):  # two launches more than slots: they are simply not timed
        vec.step(a)
    ms = eng.profile_read()
    assert len(ms) == 5 and all(0.0 < m < 50.0 for m in ms)
    assert eng.profile_read() == []          # reading starts over
    vec.step(a)
    assert len(eng.profile_read()) == 1
    eng.profile_render(0)
    vec.step(a)
    assert eng.profile_read() == []


def test_validate_state_catches_what_the_kernels_do_not_check(torch_mod):
    torch = torch_mod
    vec = _level1_vec(B=300)
    vec.reset()
    eng = vec.engine
    eng.validate(vec.puzzle_id, vec.pos)      # a reset batch is fine
    eng.validate(vec.puzzle_id)               # ids only
    ids = vec.puzzle_id.clone()
    ids[17] = vec.num_puzzles                 # one past the set
    ids[200] = -1
    with pytest.raises(ValueError, match=r"2 environment\(s\).*environment 17"):
        eng.validate(ids, vec.pos)
    pos = vec.pos.clone()
    pos[5, 0, 0] = 63                          # agent's right edge beyond the grid
    pos[9, 15, 1] = 1                          # non-zero padding slot (no Level-1 puzzle has 16 movables)
    with pytest.raises(ValueError, match=r"2 environment\(s\).*environment 5"):
        eng.validate(vec.puzzle_id, pos)
    pos[:] = vec.pos
    pos[0, 0, 1] = -1
    with pytest.raises(ValueError, match="environment 0"):
        eng.validate(vec.puzzle_id, pos)
    # every state of a random walk stays valid (legal play keeps objects inside the border walls)
    g = torch.Generator(device=vec.device).manual_seed(0)
    for _ in range(30):
        vec.step(torch.randint(0, 4, (vec.num_envs,), generator=g, device=vec.device, dtype=torch.uint8))
    eng.validate(vec.puzzle_id, vec.pos)


@pytest.mark.parametrize("kernel", ["group", "wave", "lane"])
def test_bad_actions_are_flagged_and_counted(torch_mod, kernel):
    torch = torch_mod
    vec = _level1_vec(B=256, observation=None, autoreset=False, engine_options={"step_kernel": kernel})
    vec.reset()
    before = vec.pos.clone()
    a = torch.zeros((256,), dtype=torch.uint8, device=vec.device)
    a[3], a[100], a[255] = 4, 255, 9
    vec.step(a)
    term, trunc = vec.terminated.cpu().numpy(), vec.truncated.cpu().numpy()
    bad = np.zeros(256, bool)
    bad[[3, 100, 255]] = True
    assert (term[bad] == 0xFF).all() and (trunc[bad] == 0xFF).all() and (term[~bad] != 0xFF).all()
    assert torch.equal(vec.pos[bad], before[bad])           # untouched
    assert vec.engine.bad_actions() == 3
    assert vec.engine.bad_actions() == 0                    # read-and-clear
    vec.step(torch.ones((256,), dtype=torch.uint8, device=vec.device))
    assert vec.engine.bad_actions() == 0


def test_vector_env_rejects_out_of_range_device_actions(torch_mod):
    """a device int64 action tensor was cast to uint8 unchecked (256 -> LEFT, -1 -> 255) and, with
    to_numpy=False, the 0xFF flags were never looked at."""
    torch = torch_mod
    import bench
    from pushworld_amd.vector_env import PushWorldVectorEnv

    env = PushWorldVectorEnv(bench.level1_paths()[:5], 16, max_steps=20, border_width=1, pixels_per_cell=3,
                             observation="uint8", to_numpy=False)
    env.reset(seed=1)
    dev = env.vec.device
    for bad in (256, -1, 4):
        a = torch.zeros((16,), dtype=torch.int64, device=dev)
        a[7] = bad
        with pytest.raises(ValueError, match="not in the action space"):
            env.step(a)
    env.step(torch.full((16,), 3, dtype=torch.int64, device=dev))   # in range: accepted and cast
    env.check_actions()
    a8 = torch.zeros((16,), dtype=torch.uint8, device=dev)
    a8[2] = 200                                                      # uint8 fast path: no synchronisation in step
    env.step(a8)
    with pytest.raises(ValueError, match=r"\(1 since the last check\)"):
        env.check_actions()
    env.check_actions()                                              # cleared


def test_max_steps_zero_truncates_every_step(torch_mod, golden):
    """gym_env.py:223 ``truncated = steps >= max_steps``: 0 truncates at once, only None never does (ADVICE r1:
    0 used to be encoded as "no limit")."""
    torch = torch_mod
    from oracle import pw_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    key = "bench:level1/2 Obstacle.pwp"
    oz = pw_oracle.OraclePuzzle(golden.text(key))
    for max_steps in (0, 1, None):
        vec = VecPushWorld([PushWorldPuzzle(text=golden.text(key))], 4, max_steps=max_steps, observation=None, device=0)
        vec.reset()
        oenv = pw_oracle.OracleEnv(oz, max_steps)
        oenv.reset()
        for a in (0, 2, 1, 3, 3):
            _, r, te, tr = vec.step(torch.full((4,), a, dtype=torch.uint8, device=vec.device))
            _, orew, oterm, otrunc = oenv.step(a)
            assert (tr.cpu().numpy() == int(otrunc)).all() and (te.cpu().numpy() == int(oterm)).all(), (max_steps, a)
            assert (r.cpu().numpy() == orew).all()
    with pytest.raises(ValueError):
        VecPushWorld([PushWorldPuzzle(text=golden.text(key))], 4, max_steps=-2, observation=None, device=0)


def test_entry_points_leave_the_current_device_alone(torch_mod, golden):
    """pw_engine_create / pw_search_create / novelty / upload called hipSetDevice and never
    restored it, and pw_step_render_delta allocated its scratch on whatever device was current.  With two
    devices: everything of an engine on device 1 works while the caller's current device stays 0."""
    torch = torch_mod
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.search import BreadthFirstSearch, NoveltyTables
    from pushworld_amd.vec_env import VecPushWorld

    n = torch.cuda.device_count()
    target = 1 if n >= 2 else 0
    torch.cuda.set_device(0)
    key = "bench:level1/2 Obstacle.pwp"
    pz = PushWorldPuzzle(text=golden.text(key))
    vec = VecPushWorld([pz], 64, max_steps=9, pixels_per_cell=3, border_width=1, observation="uint8", device=target,
                       autoreset=True, incremental=True)
    assert torch.cuda.current_device() == 0
    ref = VecPushWorld([pz], 64, max_steps=9, pixels_per_cell=3, border_width=1, observation="uint8", device=target,
                       autoreset=True)
    o0, o1 = vec.reset().clone(), ref.reset().clone()
    assert torch.equal(o0, o1) and torch.cuda.current_device() == 0
    g = torch.Generator(device=vec.device).manual_seed(5)
    for t in range(25):
        a = torch.randint(0, 4, (64,), generator=g, device=vec.device, dtype=torch.uint8)
        oa, ob = vec.step(a)[0], ref.step(a)[0]       # incremental path allocates its dirty-row scratch lazily
        assert torch.equal(oa, ob), t
        assert torch.cuda.current_device() == 0
    vec.engine.validate(vec.puzzle_id, vec.pos)
    assert vec.engine.bad_actions() == 0 and torch.cuda.current_device() == 0
    nt = NoveltyTables(3, 8, 8, device=target)
    nt.reset()
    assert torch.cuda.current_device() == 0
    if target == 0:
        bfs = BreadthFirstSearch(pz, max_states=4096, novelty_width=1)
        bfs.begin()
        bfs.expand()
        assert bfs.total_states > 1 and torch.cuda.current_device() == 0
        bfs.close()


@pytest.mark.parametrize("kernel", ["group", "lane", "wave"])
def test_puzzle_without_goals_terminates_at_once(torch_mod, kernel):
    """A puzzle without goals: `is_goal_state` is vacuously true (puzzle.py:409-411, all() of nothing), so every step
    terminates with reward 10 -- also after the autoreset -- exactly as the oracle's restatement does."""
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    torch = torch_mod
    text = "\n".join(["  ".join(r) for r in (["A", ".", "M0", "."], [".", ".", ".", "W"], ["M1", ".", ".", "."])])
    pz = PushWorldPuzzle(text=text)
    oz = c_oracle.COraclePuzzle(text, order="python")
    assert len(pz.goal_state) == 0 and pz.is_goal_state(pz.initial_state)
    B, T = 64, 12
    acts = np.random.default_rng(3).integers(0, 4, (T, B)).astype(np.uint8)
    pos, rew, term, trunc, steps = c_oracle.rollout_trace([oz], np.zeros(B, np.int32), acts, 5, True, 4)
    vec = VecPushWorld([pz], B, observation=None, max_steps=5, autoreset=True, device=0, engine_options={"step_kernel": kernel})
    vec.reset()
    dev = torch.as_tensor(acts).to(vec.device)
    for t in range(T):
        _, r, te, tr = vec.step(dev[t])
        assert (vec.pos.cpu().numpy() == pos[t]).all(), t
        assert (r.cpu().numpy().view(np.uint64) == rew[t].view(np.uint64)).all(), t
        assert (te.cpu().numpy() == term[t]).all() and (tr.cpu().numpy() == trunc[t]).all(), t
    assert (term[0] == 1).all() and (rew[0] == 10.0).all() and (term[1] == 0).all()  # step, autoreset, step, ...


def test_empty_batches_and_zero_steps_are_no_ops(torch_mod):
    """batch = 0 / num_steps = 0 / an empty frontier: PW_OK, nothing launched, nothing touched."""
    from pushworld_amd import _capi

    torch = torch_mod
    vec = _level1_vec(B=64, observation=None)
    vec.reset()
    before = vec.pos.clone()
    eng = vec.engine
    empty8 = torch.zeros((0,), dtype=torch.uint8, device=vec.device)
    z = lambda t: t[:0]  # noqa: E731
    eng.step(z(vec.puzzle_id), empty8, z(vec.pos), z(vec.steps), z(vec.reward), z(vec.dgoals), z(vec.terminated),
             z(vec.truncated), 0)
    vec.rollout(torch.zeros((0, 64), dtype=torch.uint8, device=vec.device))
    torch.cuda.synchronize()
    assert torch.equal(vec.pos, before) and int(vec.steps.sum()) == 0
    eng.validate(z(vec.puzzle_id), z(vec.pos))  # nothing to complain about
    eng.reset(z(vec.puzzle_id), z(vec.pos), z(vec.steps), z(vec.terminated), z(vec.truncated))
    assert _capi.lib.pw_expand4(eng.handle, 0, None, None, None, None, 0, None) == _capi.PW_OK
    assert _capi.lib.pw_expand4(eng.handle, 0, None, None, None, None, 5, None) == _capi.PW_EINVAL  # 5 states, no buffers
    assert _capi.lib.pw_step(None, None, None, None, None, None, None, None, None, 0, 0, None) == _capi.PW_EINVAL  # no engine


def test_block_order_is_chosen_from_the_batch_and_never_changes_results(torch_mod):
    """PW_OPT_STEP_BLOCK_ORDER: VecPushWorld starts the step kernel at the expensive END of a batch whose puzzles get
    heavier (more movables) towards the end; an explicit engine option wins; the order never changes a result."""
    torch = torch_mod
    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    pool = sorted((PushWorldPuzzle(p) for p in bench.level1_paths()), key=lambda z: z.num_movables)
    B = 8192
    up = (np.arange(B, dtype=np.int64) * len(pool)) // B        # light puzzles first, heavy ones last
    vecs = {
        "auto-up": VecPushWorld(pool, B, puzzle_ids=up, max_steps=30, observation=None, device=0, autoreset=True),
        "auto-down": VecPushWorld(pool, B, puzzle_ids=up[::-1].copy(), max_steps=30, observation=None, device=0, autoreset=True),
        "forced": VecPushWorld(pool, B, puzzle_ids=up, max_steps=30, observation=None, device=0, autoreset=True,
                               engine_options={"step_block_order": "forward"}),
    }
    assert vecs["auto-up"].engine.get_option("step_block_order") == 1
    assert vecs["auto-down"].engine.get_option("step_block_order") == 0
    assert vecs["forced"].engine.get_option("step_block_order") == 0
    g = torch.Generator(device="cuda:0").manual_seed(9)
    acts = torch.randint(0, 4, (40, B), generator=g, device="cuda:0", dtype=torch.uint8)
    for v in vecs.values():
        v.reset()
    for t in range(40):
        a, f = vecs["auto-up"].step(acts[t]), vecs["forced"].step(acts[t])
        d = vecs["auto-down"].step(acts[t].flip(0))
        assert torch.equal(vecs["auto-up"].pos, vecs["forced"].pos) and torch.equal(a[1], f[1]) and torch.equal(a[2], f[2])
        assert torch.equal(vecs["auto-down"].pos.flip(0), vecs["forced"].pos) and torch.equal(d[1].flip(0), f[1])
    ru, rf = vecs["auto-up"].rollout(acts), vecs["forced"].rollout(acts)
    assert torch.equal(vecs["auto-up"].pos, vecs["forced"].pos) and torch.equal(ru[0], rf[0])


def test_mixed_group_widths_on_an_unsorted_batch(torch_mod, golden):
    """N_pad 32 sets: workgroups of 32 environments run 8-lane groups unless one of their environments has more than 16
    movables (pw_step_group_mixed_kernel).  A batch in RANDOM puzzle order (wide and narrow workgroups interleaved, a ragged
    last workgroup), one step per launch and as a rollout (twin workgroups), against 16-lane groups for every environment."""
    torch = torch_mod
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("bench:")]
    pool = [PushWorldPuzzle(text=golden.text(k)) for k in keys]
    assert max(p.num_movables for p in pool) > 16
    B = 5003
    ids = np.random.default_rng(11).integers(0, len(pool), B)
    mk = lambda opts: VecPushWorld(pool, B, puzzle_ids=ids, max_steps=30, observation=None, device=0, autoreset=True,  # noqa: E731
                                   engine_options=opts)
    mixed, wide = mk({}), mk({"step_narrow_groups": 2})
    assert mixed.num_objects_padded == 32
    g = torch.Generator(device="cuda:0").manual_seed(5)
    acts = torch.randint(0, 4, (40, B), generator=g, device="cuda:0", dtype=torch.uint8)
    mixed.reset()
    wide.reset()
    for t in range(24):
        a, b = mixed.step(acts[t]), wide.step(acts[t])
        assert torch.equal(mixed.pos, wide.pos) and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:])), t
    ha, hb = mixed.rollout(acts, history=True), wide.rollout(acts, history=True)
    assert torch.equal(mixed.pos, wide.pos) and torch.equal(mixed.steps, wide.steps)
    assert all(torch.equal(x, y) for x, y in zip(ha, hb))


def test_big_state_only_batches_run_one_lane_per_environment(torch_mod):
    """PW_OPT_STEP_LANE_BATCH: a state-only launch of >= 131 072 environments (N_pad <= 16 sets) runs the one-lane-per-
    environment formulation by itself; same results as the lane groups (option "never"), step by step and as a rollout."""
    torch = torch_mod
    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    pool = [PushWorldPuzzle(p) for p in bench.level1_paths()]
    B = 196608
    ids = (np.arange(B, dtype=np.int64) * len(pool)) // B
    auto = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=25, observation=None, device=0, autoreset=True)
    groups = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=25, observation=None, device=0, autoreset=True,
                          engine_options={"step_lane_batch": "never"})
    assert auto.engine.get_option("step_lane_batch") == 131072 and groups.engine.get_option("step_lane_batch") == 2**31
    g = torch.Generator(device="cuda:0").manual_seed(4)
    acts = torch.randint(0, 4, (48, B), generator=g, device="cuda:0", dtype=torch.uint8)
    auto.reset()
    groups.reset()
    for t in range(32):
        a, b = auto.step(acts[t]), groups.step(acts[t])
        assert torch.equal(auto.pos, groups.pos) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        assert torch.equal(auto.steps, groups.steps)
    ra, rb = auto.rollout(acts), groups.rollout(acts)
    assert torch.equal(auto.pos, groups.pos) and all(torch.equal(x, y) for x, y in zip(ra, rb))
    assert int(auto.terminated.sum()) == int(groups.terminated.sum())


def test_garbage_ids_and_positions_are_memory_safe(torch_mod):
    """A third-party C caller that hands over puzzle ids outside the set or positions outside the grid gets a
    meaningless step, not a fault: ids are clamped into the set, positions are range-checked wherever they index.
    Every step / render entry point, every step kernel; the device is still healthy afterwards and a clean batch of
    the same engine is still bit-exact."""
    torch = torch_mod
    vec = _level1_vec(B=4096)
    ref = _level1_vec(B=4096)
    o_ref = ref.reset().clone()
    vec.reset()
    g = torch.Generator(device=vec.device).manual_seed(21)
    bad_ids = torch.randint(-2**31, 2**31 - 1, (vec.num_envs,), generator=g, device=vec.device, dtype=torch.int64).to(torch.int32)
    bad_pos = torch.randint(-128, 128, tuple(vec.pos.shape), generator=g, device=vec.device, dtype=torch.int16).to(torch.int8)
    good_ids = vec.puzzle_id.clone()
    eng = vec.engine
    with pytest.raises(ValueError):
        eng.validate(bad_ids, vec.pos)
    for kernel in ("group", "wave", "lane"):
        eng.set_option("step_kernel", kernel)
        for tables in ("all", "none", "big"):
            eng.set_option("step_tables", tables)
            for ids, pos in ((bad_ids, vec.pos.clone()), (good_ids, bad_pos.clone()), (bad_ids, bad_pos.clone())):
                a = torch.randint(0, 4, (vec.num_envs,), generator=g, device=vec.device, dtype=torch.uint8)
                eng.step(ids, a, pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated, vec.flags)
                eng.step_render(ids, a, pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated,
                                vec._obs_storage, vec.flags)
                eng.render(ids, pos, vec._obs_storage)
                eng.step_render_delta(ids, a, pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated,
                                      vec._obs_storage, vec.flags)
                acts = torch.randint(0, 4, (8, vec.num_envs), generator=g, device=vec.device, dtype=torch.uint8)
                eng.rollout(ids, acts, pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated, flags=vec.flags)
                eng.reset(ids, pos, vec.steps, vec.terminated, vec.truncated)
            torch.cuda.synchronize()
    eng.set_option("step_kernel", "group")
    eng.set_option("step_tables", "auto")
    # the engine and the device are fine: a clean episode equals the untouched twin
    assert torch.equal(vec.reset(), o_ref)
    for t in range(10):
        a = torch.randint(0, 4, (vec.num_envs,), generator=g, device=vec.device, dtype=torch.uint8)
        o1, r1, _, _ = vec.step(a)
        o2, r2, _, _ = ref.step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(vec.pos, ref.pos)
