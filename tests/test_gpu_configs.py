"""-m gpu: the BASELINE.json configurations as parity / property tests.

C1 single level-0 puzzle through the gym adapter; C2 4 096 copies of one level-0 puzzle,
state only; C3 65 536 Level-1 envs with render (bench workload, checked here on samples and
through size-independent properties); C4 one rank's shard (65 536 envs) of the full mix
levels 0-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _l0_text(golden, member="level0/base/train/level_0_base_train_0.pwp"):
    return golden.text("l0:" + member)


def test_c1_single_env_gym_adapter(golden, tmp_path):
    """C1: one level-0 puzzle, batch 1, default render (ppc 20, float32): 2 000 random steps
    (`numpy.random.default_rng(0)`), reset on termination; every state, reward and flag equals
    the oracle; observations are compared every 100 steps."""
    from oracle import pw_oracle
    from pushworld_amd.gym_env import PushWorldEnv

    text = _l0_text(golden)
    f = tmp_path / "c1.pwp"
    f.write_text(text)
    env = PushWorldEnv(str(f), max_steps=60)
    oz = pw_oracle.OraclePuzzle(text)
    oenv = pw_oracle.OracleEnv(oz, max_steps=60)
    rng = np.random.default_rng(0)
    obs, info = env.reset()
    ostate = oenv.reset()
    assert info["puzzle_state"] == ostate
    for t in range(2000):
        a = int(rng.integers(0, 4))
        obs, r, term, trunc, info = env.step(a)
        ostate, orew, oterm, otrunc = oenv.step(a)
        assert info["puzzle_state"] == ostate and r == orew and term == oterm and trunc == otrunc
        if t % 100 == 0:
            assert (obs == oz.observation(ostate, oz.height, oz.width)).all()
        if term or trunc:
            env.reset()
            oenv.reset()


def test_c2_4096_copies_state_only(golden):
    """C2: 4 096 copies of one level-0 puzzle, state only, max_steps 100 with autoreset: copies
    driven by the same actions stay identical and equal the oracle; copies driven by different
    actions are checked individually on a sample."""
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    text = _l0_text(golden)
    pz = PushWorldPuzzle(text=text)
    oz = c_oracle.COraclePuzzle(text)
    B, T = 4096, 400
    vec = VecPushWorld([pz], B, max_steps=100, observation=None, device=0, autoreset=True)
    vec.reset()
    gen = torch.Generator(device=vec.device)
    gen.manual_seed(0)
    acts = torch.randint(0, 4, (T, B), generator=gen, device=vec.device, dtype=torch.uint8)
    acts[:, : B // 2] = acts[:, :1]  # first half: identical action streams
    acts_h = acts.cpu().numpy()
    sample = [0, 1, B // 2 - 1, B // 2, B // 2 + 17, B - 1]
    states = {b: oz.initial_state for b in sample}
    steps = {b: 0 for b in sample}
    done = {b: False for b in sample}
    for t in range(T):
        _, r, te, tr = vec.step(acts[t])
        pos = vec.states()
        rr, tt, uu = r.cpu().numpy(), te.cpu().numpy(), tr.cpu().numpy()
        assert (pos[: B // 2] == pos[0]).all() and (rr[: B // 2] == rr[0]).all()
        for b in sample:
            if done[b]:  # next-step autoreset
                states[b], steps[b], done[b] = oz.initial_state, 0, False
                want = (states[b], 0.0, False, False)
            else:
                s, rew, term = oz.env_step(states[b], int(acts_h[t, b]))
                steps[b] += 1
                trunc = steps[b] >= 100
                states[b], done[b] = s, term or trunc
                want = (s, rew, term, trunc)
            assert tuple(map(tuple, pos[b, : oz.num_movables].tolist())) == want[0], (t, b)
            assert rr[b] == want[1] and bool(tt[b]) == want[2] and bool(uu[b]) == want[3], (t, b)


@pytest.mark.parametrize("fused", [False, True])
def test_c3_level1_mix_full_batch(golden, fused):
    """C3 at the bench's full size (65 536 envs, 68 Level-1 puzzles, frame 51x42, uint8 ppc 3):
    sampled envs equal the oracle (state, reward, flags, full observation); all envs satisfy the
    size-independent properties: envs of one puzzle with equal action histories are identical;
    the observation is a pure function of (puzzle, state) [re-render == step output];
    padding bytes are zero; step counters advance by one."""
    import torch

    import bench
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    B, T = 65536, 12
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                       border_width=1, observation="uint8", device=0, autoreset=True, fused=fused)
    assert vec.engine.obs_shape == (153, 126, 3)
    vec.reset()
    gen = torch.Generator(device=vec.device)
    gen.manual_seed(1)
    acts = torch.randint(0, 4, (T, B), generator=gen, device=vec.device, dtype=torch.uint8)
    acts[:, 1::2] = acts[:, 0::2]  # neighbours (same puzzle except at group borders) share actions
    acts_h = acts.cpu().numpy()
    sample = list(range(0, B, 1543)) + [B - 1]
    oracles = {}
    for b in sample:
        if ids[b] not in oracles:
            with open(paths[ids[b]]) as f:
                oracles[ids[b]] = c_oracle.COraclePuzzle(f.read())
    states = {b: oracles[ids[b]].initial_state for b in sample}
    for t in range(T):
        obs, r, te, tr = vec.step(acts[t])
        pos = vec.states()
        same = ids[0::2] == ids[1::2]
        assert (pos[0::2][same] == pos[1::2][same]).all()
        assert (vec.steps.cpu().numpy() == t + 1).all()
        rr, tt = r.cpu().numpy(), te.cpu().numpy()
        for b in sample:
            oz = oracles[ids[b]]
            s, rew, term = oz.env_step(states[b], int(acts_h[t, b]))
            states[b] = s
            assert tuple(map(tuple, pos[b, : oz.num_movables].tolist())) == s and rr[b] == rew and bool(tt[b]) == term
        if t in (0, T - 1):
            first = obs.clone()
            again = vec.render()
            assert torch.equal(first, again)
            for b in sample[::4]:
                oz = oracles[ids[b]]
                assert (obs[b].cpu().numpy() == oz.observation(states[b], 51, 42, 3, 1, dtype="u8")).all(), b
            # zero padding around every puzzle smaller than the frame (env_utils.py:75-91)
            b = sample[3]
            oz = oracles[ids[b]]
            top = (153 - oz.height * 3) // 2
            left = (126 - oz.width * 3) // 2
            img = obs[b].cpu().numpy()
            assert img[:top].sum() == 0 and img[:, :left].sum() == 0
            assert img[top + oz.height * 3:].sum() == 0 and img[:, left + oz.width * 3:].sum() == 0


def test_c4_full_mix_shard(golden):
    """C4, one rank's shard: 65 536 envs, 50 % level 0 (7 families, train) + 50 % levels 1-4
    (223 puzzles), frame 54x47, NP 32, state-only headline + uint8 ppc 3 render variant.
    Sampled envs equal the oracle over a random walk; observations of sampled envs equal the
    oracle's padded image."""
    import torch

    from oracle import c_oracle
    from pushworld_amd.benchmark_data import level0_texts, level_paths
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    l0 = level0_texts(limit=300)  # 7 x 300 level-0 puzzles
    texts = list(l0.values())
    for lv in (1, 2, 3, 4):
        for p in level_paths(lv):
            with open(p) as f:
                texts.append(f.read())
    n0 = len(l0)
    pool = [PushWorldPuzzle(text=t) for t in texts]
    B, T = 65536, 40
    rng = np.random.default_rng(100)
    ids = np.concatenate([rng.integers(0, n0, B // 2), rng.integers(n0, len(pool), B // 2)])
    ids.sort()
    vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=None, pixels_per_cell=3, border_width=1,
                       observation="uint8", pad_cells=(54, 47), device=0)
    assert vec.num_objects_padded == 32 and vec.engine.obs_shape == (162, 141, 3)
    vec.reset()
    gen = torch.Generator(device=vec.device)
    gen.manual_seed(100)
    acts = torch.randint(0, 4, (T, B), generator=gen, device=vec.device, dtype=torch.uint8)
    acts_h = acts.cpu().numpy()
    sample = list(range(0, B, 997))
    oracles = {i: c_oracle.COraclePuzzle(texts[i]) for i in {int(ids[b]) for b in sample}}
    states = {b: oracles[int(ids[b])].initial_state for b in sample}
    for t in range(T):
        obs, r, te, tr = vec.step(acts[t])
        if t % 13 and t != T - 1:
            for b in sample:
                states[b] = oracles[int(ids[b])].env_step(states[b], int(acts_h[t, b]))[0]
            continue
        pos, rr, tt = vec.states(), r.cpu().numpy(), te.cpu().numpy()
        for k, b in enumerate(sample):
            oz = oracles[int(ids[b])]
            s, rew, term = oz.env_step(states[b], int(acts_h[t, b]))
            states[b] = s
            assert tuple(map(tuple, pos[b, : oz.num_movables].tolist())) == s and rr[b] == rew and bool(tt[b]) == term
            if k % 8 == 0:
                assert (obs[b].cpu().numpy() == oz.observation(s, 54, 47, 3, 1, dtype="u8")).all(), (t, b)


@pytest.mark.parametrize("ppc,obs", [(3, "uint8"), (4, "uint8"), (3, "float32")])
def test_small_frames_many_envs(golden, ppc, obs):
    """Level-0 puzzles in their own (small) frame: observations of 1.3-3.9 KB, i.e. several
    environments per 4 KiB page of the buffer.  Every environment's observation equals the
    oracle's along a random walk."""
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("l0:")][::6]
    texts = [golden.text(k) for k in keys]
    pool = [PushWorldPuzzle(text=t) for t in texts]
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    B = 3 * len(pool)
    ids = np.arange(B) % len(pool)
    vec = VecPushWorld(pool, B, puzzle_ids=ids, pixels_per_cell=ppc, border_width=1, observation=obs, device=0)
    fh, fw = vec.engine.obs_shape[0] // ppc, vec.engine.obs_shape[1] // ppc
    assert vec.engine.obs_bytes < 4096 * (4 if obs == "float32" else 1) * (2 if ppc == 4 else 1)
    vec.reset()
    gen = torch.Generator(device=vec.device)
    gen.manual_seed(9)
    acts = torch.randint(0, 4, (30, B), generator=gen, device=vec.device, dtype=torch.uint8)
    acts_h = acts.cpu().numpy()
    states = [oracles[i].initial_state for i in ids]
    for t in range(30):
        o = vec.step(acts[t])[0]
        for b in range(B):
            states[b] = oracles[ids[b]].get_next_state(states[b], int(acts_h[t, b]))
        if t % 7 == 0 or t == 29:
            img = o.cpu().numpy()
            for b in range(B):
                want = oracles[ids[b]].observation(states[b], fh, fw, ppc, 1, dtype="u8" if obs == "uint8" else "f32")
                assert (img[b] == want).all(), (t, b)


def test_frames_just_above_one_page(golden):
    """Observation of 5.8 KB (frame 18x12 cells): every 4 KiB page of the buffer straddles two
    environments, the code path of the page-ordered render kernel that the C3 frame exercises
    only once per environment."""
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("pytest:")]
    texts = [golden.text(k) for k in keys]
    pool = [PushWorldPuzzle(text=t) for t in texts]
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    B = 37 * len(pool)
    ids = np.sort(np.arange(B) % len(pool))
    vec = VecPushWorld(pool, B, puzzle_ids=ids, pixels_per_cell=3, border_width=1, observation="uint8", device=0)
    assert vec.engine.render_kernel == "pw_render_page_kernel" and vec.engine.obs_shape == (54, 36, 3)
    vec.reset()
    gen = torch.Generator(device=vec.device)
    gen.manual_seed(11)
    acts = torch.randint(0, 4, (40, B), generator=gen, device=vec.device, dtype=torch.uint8)
    acts_h = acts.cpu().numpy()
    states = [oracles[i].initial_state for i in ids]
    for t in range(40):
        o = vec.step(acts[t])[0]
        for b in range(B):
            states[b] = oracles[ids[b]].get_next_state(states[b], int(acts_h[t, b]))
        if t % 13 == 0 or t == 39:
            img = o.cpu().numpy()
            for b in range(B):
                assert (img[b] == oracles[ids[b]].observation(states[b], 18, 12, 3, 1, dtype="u8")).all(), (t, b)


def test_engine_from_packed_set_file(golden, tmp_path):
    """a VecPushWorld built from PuzzleSet.load(<packed file>) steps and renders exactly
    like the one built from the puzzle texts."""
    import torch
    from pushworld_amd import _capi
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith(("bench:level1/", "rand:"))][::5]
    pool = [PushWorldPuzzle(text=golden.text(k)) for k in keys]
    B, T = 512, 40
    a = VecPushWorld(pool, B, max_steps=25, pixels_per_cell=3, border_width=1, observation="uint8", autoreset=True)
    path = str(tmp_path / "pool.pwset")
    a.pset.save(path)
    b = VecPushWorld(_capi.PuzzleSet.load(path, a.device.index), B, max_steps=25, pixels_per_cell=3, border_width=1,
                     observation="uint8", autoreset=True)
    assert b.puzzles is None and b.num_puzzles == len(pool) and b.engine.obs_shape == a.engine.obs_shape
    acts = torch.randint(0, 4, (T, B), dtype=torch.uint8, device=a.device, generator=torch.Generator(device=a.device).manual_seed(3))
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    for t in range(T):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), t
    assert torch.equal(a.pos, b.pos)


def test_batch_beyond_2_31_chunks():
    """600 000 environments at the C3 frame: a 34.7 GB observation buffer = 2.17e9 16-byte chunks, more than
    the page kernel numbers with 32 bits, so the render runs in slices.  Every environment gets the same
    actions; environments of one puzzle must then equal the corresponding environment of a small batch."""
    import torch

    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    pool = [PushWorldPuzzle(p) for p in paths]
    B = 600_000
    ids = (np.arange(B, dtype=np.int64) * len(pool)) // B
    big = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=50, pixels_per_cell=3, border_width=1, observation="uint8",
                       autoreset=True)
    small = VecPushWorld(pool, len(pool), puzzle_ids=np.arange(len(pool)), max_steps=50, pixels_per_cell=3, border_width=1,
                         observation="uint8", autoreset=True)
    assert big.engine.render_kernel == "pw_render_page_kernel"
    big.reset()
    small.reset()
    probe = torch.as_tensor(np.concatenate([np.arange(0, B, 7919), [B - 1, B - 2, 371_085, 371_086, 371_087]]), device=big.device)
    rng = np.random.default_rng(0)
    for t in range(6):
        a = int(rng.integers(0, 4))
        ob, _, _, _ = big.step(torch.full((B,), a, dtype=torch.uint8, device=big.device))
        os_, _, _, _ = small.step(torch.full((len(pool),), a, dtype=torch.uint8, device=small.device))
        want = os_[big.puzzle_id[probe].long()]
        assert torch.equal(ob[probe], want), t
    # the bytes between observations (stride padding) stay untouched (zero) everywhere, also at slice boundaries
    pad = big._obs_storage[:, big.engine.obs_bytes:]
    assert int(pad.max()) == 0
    del big, ob
    torch.cuda.empty_cache()


def _compare_all_envs(vec, oracles, ids, acts, max_steps, frame, n_obs, obs_steps, ppc=3, bw=1, dtype="u8"):
    """Every environment, every step: positions, float64 reward bits, terminated, truncated, step counter
    against the C oracle's trace (OpenMP over environments, next-step autoreset); full observations of
    ``n_obs`` environments spread over the batch at the steps ``obs_steps`` (``ppc`` / ``bw`` / ``dtype``: the
    engine's observation settings)."""
    import torch

    from oracle import c_oracle

    T, B = acts.shape
    NP = vec.num_objects_padded
    want_pos, want_r, want_te, want_tr, want_steps = c_oracle.rollout_trace(oracles, ids, acts, max_steps, True, NP)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    sel = np.unique(np.linspace(0, B - 1, n_obs).astype(np.int64))
    assert vec.reset() is not None
    resets = 0
    for t in range(T):
        obs, r, te, tr = vec.step(acts_dev[t])
        pos = vec.pos.cpu().numpy()
        bad = np.nonzero((pos != want_pos[t]).any(axis=(1, 2)))[0]
        assert bad.size == 0, (t, bad[:5], pos[bad[:1]], want_pos[t][bad[:1]])
        assert (r.cpu().numpy().view(np.uint64) == want_r[t].view(np.uint64)).all(), t
        assert (te.cpu().numpy() == want_te[t]).all() and (tr.cpu().numpy() == want_tr[t]).all(), t
        assert (vec.steps.cpu().numpy() == want_steps[t]).all(), t
        resets += int((want_steps[t] == 0).sum())
        if t in obs_steps:
            got = obs[torch.as_tensor(sel).to(vec.device)].cpu().numpy()
            want = c_oracle.observe_batch(oracles, ids, want_pos[t], sel, frame[0], frame[1], ppc, bw, dtype)
            assert got.dtype == want.dtype and got.shape == want.shape
            diff = np.nonzero((got != want).any(axis=(1, 2, 3)))[0]
            assert diff.size == 0, (t, sel[diff[:5]])
    return resets, len(sel)


@pytest.mark.parametrize("obs,ppc,bw,B,n_obs,kernel", [
    ("float32", 20, 2, 2048, 256, "pw_render_rowpage_kernel"),  # the reference's default observation, 10.3 MB each
    ("uint8", 20, 2, 4096, 256, "pw_render_rowpage_kernel"),    # 2 520-byte rows: the unaligned instance
    ("uint8", 8, 2, 16384, 384, "pw_render_rowpage_kernel"),
    ("float32", 3, 1, 32768, 512, "pw_render_page_kernel"),     # the float32 instance of the ppc-3 page kernel
])
def test_row_page_kernels_against_the_oracle_at_batch_scale(obs, ppc, bw, B, n_obs, kernel):
    """The page-ordered kernels of every setting other than uint8 / ppc 3, on the Level-1 mix at batch scale, pinned to
    the ORACLE directly (puzzle.py:596-638, env_utils.py:65-91): every environment's state at every step with next-step
    autoresets inside the window, and >= 256 complete observations per setting at three steps; then the same
    observations for in-bounds OVERLAPPING states (every object at the position of its neighbour in the state vector:
    painter order, puzzle.py:453-458)."""
    import torch

    import bench
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    texts = [open(p).read() for p in paths]
    T, max_steps = 8, 5
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=max_steps,
                       pixels_per_cell=ppc, border_width=bw, observation=obs, device=0, autoreset=True, fused=True)
    assert vec.engine.obs_shape == (51 * ppc, 42 * ppc, 3) and vec.engine.render_kernel == kernel
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    acts = np.random.default_rng(17 + ppc).integers(0, 4, size=(T, B), dtype=np.uint8)
    dt = "f32" if obs == "float32" else "u8"
    resets, n_sel = _compare_all_envs(vec, oracles, ids, acts, max_steps, (51, 42), n_obs, (0, 5, T - 1), ppc, bw, dt)
    assert resets >= B and n_sel >= min(n_obs, 256)
    # overlapping states: object j at the position of object j - 1 (clamped into the grid by its own bounding box)
    pos = vec.states()
    over = pos.copy()
    for b in range(B):
        pz = oracles[int(ids[b])]
        n = pz.num_movables
        for j in range(1, n):
            w = max(c[0] for c in pz.py.shapes[j]) + 1
            h = max(c[1] for c in pz.py.shapes[j]) + 1
            over[b, j, 0] = min(int(pos[b, j - 1, 0]), pz.width - w)
            over[b, j, 1] = min(int(pos[b, j - 1, 1]), pz.height - h)
    vec.set_states(over)
    got_all = vec.render()
    sel = np.unique(np.linspace(0, B - 1, min(n_obs, 256)).astype(np.int64))
    got = got_all[torch.as_tensor(sel).to(vec.device)].cpu().numpy()
    want = c_oracle.observe_batch(oracles, ids, over, sel, 51, 42, ppc, bw, dt)
    diff = np.nonzero((got != want).any(axis=(1, 2, 3)))[0]
    assert diff.size == 0, sel[diff[:5]]


def test_c3_every_environment_against_the_oracle():
    """C3 at full size: ALL 65 536 environments, 10 steps with max_steps 6 (so thousands of next-step autoresets
    happen inside the window): state, float64 reward, flags and step counter of every environment at every step
    equal the C oracle's; 1 024 complete observations (uint8 ppc 3, frame 51 x 42) at three of the steps."""
    import bench
    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    texts = [open(p).read() for p in paths]
    B, T, max_steps = 65536, 10, 6
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    vec = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, puzzle_ids=ids, max_steps=max_steps,
                       pixels_per_cell=3, border_width=1, observation="uint8", device=0, autoreset=True, fused=True)
    assert vec.engine.obs_shape == (153, 126, 3) and vec.engine.render_kernel == "pw_render_page_kernel"
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    acts = np.random.default_rng(3).integers(0, 4, size=(T, B), dtype=np.uint8)
    resets, n_obs = _compare_all_envs(vec, oracles, ids, acts, max_steps, (51, 42), 1024, (0, 6, T - 1))
    assert resets >= B and n_obs >= 1000   # every environment was truncated and reset at least once


@pytest.mark.parametrize("rank", [0, 5])
def test_c4_shard_with_the_full_level0_pool_against_the_oracle(rank):
    """C4 as specified: the rank's 65 536-environment shard of the 524 288-environment assignment
    over ALL 14 000 Level-0 train puzzles + the 223 puzzles of Levels 1-4 (N_pad 32, frame 54 x 47), built the
    way ``bench.py --config c4`` builds it: every environment at every step against the C oracle, and 1 024
    complete uint8 ppc-3 observations."""
    from oracle import c_oracle
    from pushworld_amd import _capi
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.sharding import c4_global_puzzle_ids, shard_puzzle_ids
    from pushworld_amd.vec_env import VecPushWorld

    texts = list(bd.level0_texts().values())
    n_l0 = len(texts)
    for lv in (1, 2, 3, 4):
        for p in bd.level_paths(lv):
            with open(p) as f:
                texts.append(f.read())
    assert n_l0 == 14000 and len(texts) == 14223
    B, T, max_steps = 65536, 9, 5
    ids = np.sort(shard_puzzle_ids(c4_global_puzzle_ids(8 * B, n_l0, len(texts) - n_l0, 100), rank, 8))
    pset = _capi.PuzzleSet([_capi.ParsedPuzzle(t) for t in texts], 0)
    vec = VecPushWorld(pset, B, puzzle_ids=ids, max_steps=max_steps, pixels_per_cell=3, border_width=1,
                       observation="uint8", pad_cells=(54, 47), device=0, autoreset=True, fused=True)
    assert vec.num_objects_padded == 32 and vec.engine.obs_shape == (162, 141, 3)
    used = np.unique(ids)
    remap = np.full(len(texts), -1, np.int64)
    remap[used] = np.arange(len(used))
    oracles = [c_oracle.COraclePuzzle(texts[int(p)]) for p in used]
    acts = np.random.default_rng(100 + rank).integers(0, 4, size=(T, B), dtype=np.uint8)
    # the oracle sees a compacted pool (only the puzzles of this shard), the engine the full one
    import torch
    T_, B_ = acts.shape
    want = c_oracle.rollout_trace(oracles, remap[ids], acts, max_steps, True, 32)
    acts_dev = torch.as_tensor(acts).to(vec.device)
    sel = np.unique(np.linspace(0, B - 1, 1024).astype(np.int64))
    vec.reset()
    for t in range(T):
        obs, r, te, tr = vec.step(acts_dev[t])
        pos = vec.pos.cpu().numpy()
        bad = np.nonzero((pos != want[0][t]).any(axis=(1, 2)))[0]
        assert bad.size == 0, (t, bad[:5])
        assert (r.cpu().numpy().view(np.uint64) == want[1][t].view(np.uint64)).all(), t
        assert (te.cpu().numpy() == want[2][t]).all() and (tr.cpu().numpy() == want[3][t]).all(), t
        assert (vec.steps.cpu().numpy() == want[4][t]).all(), t
        if t in (0, 5, T - 1):
            got = obs[torch.as_tensor(sel).to(vec.device)].cpu().numpy()
            ref = c_oracle.observe_batch(oracles, remap[ids], want[0][t], sel, 54, 47, 3, 1)
            diff = np.nonzero((got != ref).any(axis=(1, 2, 3)))[0]
            assert diff.size == 0, (t, sel[diff[:5]])
    assert (want[4] == 0).sum() >= B
