"""pw_step_render_delta (incremental observation maintenance) against the full render: every byte of
the observation buffer, every step, under autoreset, puzzle re-sampling, illegal overlapping
states, several frames.  The full render itself is pinned to the reference in test_gpu_parity.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pool(golden, keys):
    from pushworld_amd.puzzle import PushWorldPuzzle

    return [PushWorldPuzzle(text=golden.text(k)) for k in keys]


@pytest.mark.parametrize("pad,resample", [(None, False), ((54, 47), False), (None, True)])
def test_incremental_equals_full_render_level1(golden, pad, resample):
    import torch
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("bench:level1/")]
    pool = _pool(golden, keys)
    B, T = 1536, 130
    ids = (np.arange(B) * len(pool)) // B
    kw = dict(puzzle_ids=ids, max_steps=35, pixels_per_cell=3, border_width=1, observation="uint8", pad_cells=pad,
              autoreset=True, resample=resample, seed=11)
    full = VecPushWorld(pool, B, **kw)
    inc = VecPushWorld(pool, B, incremental=True, **kw)
    assert inc.engine.render_kernel in ("pw_render_page_kernel", "pw_render_u8_ppc3_kernel")
    g = torch.Generator(device=full.device).manual_seed(2)
    acts = torch.randint(0, 4, (T, B), dtype=torch.uint8, device=full.device, generator=g)
    assert torch.equal(full.reset(seed=11), inc.reset(seed=11))
    n_reset = 0
    for t in range(T):
        done_before = (full.terminated | full.truncated) != 0
        n_reset += int(done_before.sum())
        fo = full.step(acts[t])
        io = inc.step(acts[t])
        for x, y in zip(fo, io):
            assert torch.equal(x, y), t
        assert torch.equal(full.pos, inc.pos) and torch.equal(full.puzzle_id, inc.puzzle_id), t
        assert torch.equal(full._obs_storage, inc._obs_storage), t  # including the stride padding bytes
    assert n_reset > B  # every env went through at least one autoreset on average


def test_incremental_handles_overlapping_states_and_small_frames(golden):
    """Random in-bounds states where objects overlap each other, walls and goals (the states of the
    'not already overlapping' parity tests): first step renders fully, the following ones incrementally."""
    import torch
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith(("pytest:", "rand:"))][:60]
    pool = _pool(golden, keys)
    B = len(pool) * 8
    ids = np.arange(B) % len(pool)
    rng = np.random.default_rng(5)
    full = VecPushWorld(pool, B, puzzle_ids=ids, pixels_per_cell=3, border_width=1, observation="uint8")
    inc = VecPushWorld(pool, B, puzzle_ids=ids, pixels_per_cell=3, border_width=1, observation="uint8", incremental=True)
    full.reset()
    inc.reset()
    pos = np.zeros((B, full.engine.np, 2), np.int8)
    for b in range(B):
        p = pool[ids[b]]
        w, h = p.dimensions
        for j in range(p.num_movables):
            pos[b, j] = (rng.integers(1, max(2, w - 1)), rng.integers(1, max(2, h - 1)))
    full.set_states(pos)
    inc.set_states(pos)
    g = torch.Generator(device=full.device).manual_seed(9)
    for t in range(40):
        a = torch.randint(0, 4, (B,), dtype=torch.uint8, device=full.device, generator=g)
        fo, io = full.step(a), inc.step(a)
        assert torch.equal(fo[0], io[0]), t
        assert torch.equal(full.pos, inc.pos), t
        if t == 20:  # an external state change invalidates the buffer: next step falls back to the full path
            inc.set_states(full.states())
            full.set_states(full.states())


@pytest.mark.parametrize("kw", [dict(observation="float32", pixels_per_cell=3, border_width=1),
                                dict(observation="uint8", pixels_per_cell=8, border_width=2),
                                dict(observation="float32", pixels_per_cell=5, border_width=2),
                                dict(observation="uint8", pixels_per_cell=20, border_width=2)])
def test_incremental_generic_kernel_equals_full_render(golden, kw):
    """Every engine that is not uint8 / ppc 3 redraws the changed cell rows with the generic LDS kernel
    (float32, other pixel sizes incl. the reference default 20 / 2) while its full render is a page-ordered
    kernel: identical buffers every step, with
    autoreset + re-sampling over a mixed pool (different puzzle heights -> whole-frame redraws)."""
    import torch
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("bench:level1/")][::3] + \
           [k for k in golden.keys if k.startswith(("pytest:", "rand:"))][:20]
    pool = _pool(golden, keys)
    B, T = 3 * len(pool), 70
    ids = np.arange(B) % len(pool)
    common = dict(puzzle_ids=ids, max_steps=19, autoreset=True, resample=True, seed=21, **kw)
    full = VecPushWorld(pool, B, **common)
    inc = VecPushWorld(pool, B, incremental=True, **common)
    # the FULL render of these engines is page-ordered (ppc-3 page kernel / row-page kernel, or the LDS kernel for rows
    # shorter than 512 bytes); the incremental redraw of the changed cell rows is the generic LDS kernel in every case
    if (kw["observation"], kw["pixels_per_cell"]) == ("float32", 3):
        assert inc.engine.render_kernel == "pw_render_page_kernel"
    else:
        assert inc.engine.render_kernel in ("pw_render_rowpage_kernel", "pw_render_generic_kernel")
    g = torch.Generator(device=full.device).manual_seed(1)
    assert torch.equal(full.reset(seed=21), inc.reset(seed=21))
    for t in range(T):
        a = torch.randint(0, 4, (B,), dtype=torch.uint8, device=full.device, generator=g)
        fo, io = full.step(a), inc.step(a)
        assert torch.equal(fo[0], io[0]), t
        assert torch.equal(full._obs_storage, inc._obs_storage), t
        assert torch.equal(full.pos, inc.pos) and torch.equal(fo[1], io[1]), t
    # overlapping states: painter order inside the redrawn rows
    pos = full.states()
    pos[:, 1:] = pos[:, :-1]
    full.set_states(pos)
    inc.set_states(pos)
    for t in range(6):
        a = torch.randint(0, 4, (B,), dtype=torch.uint8, device=full.device, generator=g)
        assert torch.equal(full.step(a)[0], inc.step(a)[0]), t


@pytest.mark.parametrize("cfg", [None, (0, 0, 0), (1, 0, 7), (2, 4, 3)])
def test_incremental_with_large_pool_static_images_in_hbm(golden, cfg):
    """860 puzzles of every size in a 64 x 64 frame: 95 MB of static images, of which the page-ordered kernel reads
    only each puzzle's own pixel rows (the frame padding above and below is neither loaded nor computed).  The
    page-ordered full render, the per-environment LDS kernel and the incremental path stay byte-identical, through
    autoreset with re-sampling (environments of one puzzle no longer sit next to each other)."""
    import torch
    from pushworld_amd.vec_env import VecPushWorld

    pool = _pool(golden, list(golden.keys))
    B, T = 2 * len(pool), 60
    ids = np.arange(B) % len(pool)
    kw = dict(puzzle_ids=ids, max_steps=17, pixels_per_cell=3, border_width=1, observation="uint8", pad_cells=(64, 64),
              autoreset=True, resample=True, seed=3)
    opts = {} if cfg is None else dict(zip(("page_order", "page_run_log2", "page_lds_pad_kb"), cfg))
    full = VecPushWorld(pool, B, engine_options=opts, **kw)
    lds = VecPushWorld(pool, B, engine_options={"render_kernel": "lds"}, **kw)
    inc = VecPushWorld(pool, B, incremental=True, **kw)
    assert full.engine.render_kernel == "pw_render_page_kernel" and lds.engine.render_kernel == "pw_render_u8_ppc3_kernel"
    g = torch.Generator(device=full.device).manual_seed(4)
    first = full.reset(seed=3)
    assert torch.equal(first, inc.reset(seed=3)) and torch.equal(first, lds.reset(seed=3))
    for t in range(T):
        a = torch.randint(0, 4, (B,), dtype=torch.uint8, device=full.device, generator=g)
        fo, lo, io = full.step(a), lds.step(a), inc.step(a)
        assert torch.equal(fo[0], io[0]) and torch.equal(fo[0], lo[0]), t
        assert torch.equal(full.puzzle_id, inc.puzzle_id) and torch.equal(full.pos, inc.pos), t


def test_incremental_full_size_is_a_pure_function_of_the_state():
    """C3 at the bench's full size (65 536 envs): after 60 incremental steps with autoreset the buffer equals
    a from-scratch render of the final states, byte for byte, for the whole batch (observation = pure
    function of (puzzle, state)); a second pass with max_steps 7 forces thousands of resets per step."""
    import torch

    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    B = 65536
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    for max_steps, T in ((200, 60), (7, 25)):
        vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=max_steps, pixels_per_cell=3,
                           border_width=1, observation="uint8", autoreset=True, incremental=True)
        vec.reset()
        g = torch.Generator(device=vec.device).manual_seed(max_steps)
        for t in range(T):
            vec.step(torch.randint(0, 4, (B,), dtype=torch.uint8, device=vec.device, generator=g))
        got = vec._obs_storage.clone()
        vec.render()  # full render of the same states into the same buffer
        assert torch.equal(got, vec._obs_storage)
        del vec, got
        torch.cuda.empty_cache()


@pytest.mark.parametrize("kw", [dict(observation="float32", pixels_per_cell=20, border_width=2),   # the gym default: 8 workgroups
                                dict(observation="uint8", pixels_per_cell=8, border_width=2),      # a small frame: one workgroup
                                dict(observation="float32", pixels_per_cell=5, border_width=1),
                                dict(observation="uint8", pixels_per_cell=20, border_width=3)])    # uint8 frames of >= 64 KB: the one-launch form in uint8
@pytest.mark.parametrize("fused", [1, 2, 0])  # PW_OPT_STEP_ONE_FUSED: one launch writing the changed columns / whole rows, two launches
def test_batch_of_one_completion_word_and_split_redraw(golden, kw, fused):
    """pw_step_render_delta on a batch of ONE (what the gym / dm_env adapters launch): the changed rows leave through eight
    workgroups where the frame has >= 64 KB (since round 6 in the same launch as the step: the call returns 2, and only the changed
    COLUMNS of those rows are written), the call returns 1 / 2 and the last workgroup writes the call's number into the engine's
    completion word (pw_engine_set_step_signal) after everything else -- polled here WITHOUT a stream synchronisation; the buffer
    equals the full render's every step, through autoresets (whole-frame redraws) and blocked moves (nothing to redraw)."""
    import time

    import torch
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("bench:level1/")][:3]
    for key in keys:
        pool = _pool(golden, [key])
        common = dict(max_steps=14, autoreset=True, **kw)
        full = VecPushWorld(pool, 1, **common)
        inc = VecPushWorld(pool, 1, incremental=True, **common)
        assert inc.engine.get_option("step_one_fused") == 1
        inc.engine.set_option("step_one_fused", fused)
        word = torch.zeros((1,), dtype=torch.int64).pin_memory()
        inc.engine.set_step_signal(word)
        assert torch.equal(full.reset(), inc.reset())
        torch.cuda.synchronize()
        g = torch.Generator(device=full.device).manual_seed(4)
        signalled = 0
        for t in range(60):
            a = torch.randint(0, 4, (1,), dtype=torch.uint8, device=full.device, generator=g)
            torch.cuda.synchronize()  # (the action is there; nothing below waits for the stream)
            rc = inc._call_step_delta(a.data_ptr())
            # the word will be written -- by the redraw kernel behind the step kernel (1), or by the ONE launch that does both (2:
            # round 6, frames of at least 64 KiB -- state in device memory, autoresets and blocked moves included)
            assert rc == (2 if fused and int(inc.engine.obs_bytes) >= (64 << 10) else 1), (key, t, rc)
            signalled += 1
            t0 = time.time()
            while int(word[0]) != signalled:
                assert time.time() - t0 < 5.0, (key, t, int(word[0]))
            got = inc._obs_storage.clone()  # (queued behind the kernels that have already reported)
            fo = full.step(a)
            assert torch.equal(full._obs_storage, got), (key, t)
            assert torch.equal(full.pos, inc.pos) and torch.equal(fo[1], inc.reward) and torch.equal(fo[2], inc.terminated), (key, t)
        inc.engine.set_step_signal(None)
        a = torch.zeros((1,), dtype=torch.uint8, device=full.device)
        assert inc._call_step_delta(a.data_ptr()) == 0  # switched off: PW_OK, no word
        full.step(a)
        torch.cuda.synchronize()
        assert torch.equal(full._obs_storage, inc._obs_storage) and int(word[0]) == signalled
