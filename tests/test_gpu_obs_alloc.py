"""-m gpu: observation buffers owned by the library (pw_obs_alloc / pw_obs_alloc_tuned / pw_obs_free): torch adopts
them in place, the observations written into them are those of the oracle, memory goes back to the DEVICE (not to
torch's caching allocator), and VecPushWorld binds its observation tensor once, in the constructor."""
import ctypes
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _level1(B, **kw):
    import bench
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    paths = bench.level1_paths()
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    args = dict(puzzle_ids=ids, max_steps=50, pixels_per_cell=3, border_width=1, observation="uint8", device=0,
                autoreset=True)
    args.update(kw)
    texts = [open(p).read() for p in paths]
    return VecPushWorld([PushWorldPuzzle(p) for p in paths], B, **args), texts, ids


def _oracle_obs(texts, ids, pos, envs, pad_h, pad_w):
    from oracle import c_oracle

    puzzles = [c_oracle.COraclePuzzle(t) for t in texts]
    return c_oracle.observe_batch(puzzles, ids, pos, envs, pad_h, pad_w, 3, 1, "u8")


def test_owned_buffer_is_adopted_in_place_and_freed_to_the_device(torch_mod):
    torch = torch_mod
    vec, texts, ids = _level1(2048, tune=False)
    ref = vec.reset().clone()
    eng = vec.engine
    torch.cuda.synchronize()
    for cycle in range(3):
        reserved0 = torch.cuda.memory_reserved()
        st, view = eng.alloc_obs_owned(vec.num_envs)
        assert st.device == vec.device and st.dtype == torch.uint8 and tuple(view.shape) == tuple(ref.shape)
        assert torch.cuda.memory_reserved() == reserved0  # the memory is the device's, not torch's allocator's
        assert int(st.max()) == 0  # zero-filled
        eng.render(vec.puzzle_id, vec.pos, st)
        torch.cuda.synchronize()
        assert torch.equal(view, ref)
        sub = view[3]  # a view keeps the buffer alive
        del st, view
        gc.collect()
        assert torch.equal(sub, ref[3])
        del sub
        gc.collect()
    # released to the DEVICE: 40 buffers of 8.7 GB one after the other are more than the 288 GB of HBM
    reserved0 = torch.cuda.memory_reserved()
    for cycle in range(40):
        st, view = eng.alloc_obs_owned(150000)
        st[-1].fill_(cycle)
        assert int(st[-1, -1]) == cycle and int(st[0, 0]) == 0
        del st, view
    assert torch.cuda.memory_reserved() == reserved0
    # against the oracle, not only against the torch-owned buffer
    pos = vec.states()
    envs = [0, 1, 700, 2047]
    want = _oracle_obs(texts, ids, pos, envs, ref.shape[1] // 3, ref.shape[2] // 3)
    assert np.array_equal(ref[envs].cpu().numpy(), want)


def test_obs_free_rejects_foreign_pointers(torch_mod):
    from pushworld_amd import _capi

    vec, _, _ = _level1(256, tune=False)
    with pytest.raises(ValueError):
        _capi.check(_capi.lib.pw_obs_free(vec.engine.handle, ctypes.c_void_p(vec._obs_storage.data_ptr())))
    assert _capi.lib.pw_obs_free(vec.engine.handle, None) == 0
    p = ctypes.c_void_p()
    with pytest.raises(ValueError):
        _capi.check(_capi.lib.pw_obs_alloc(vec.engine.handle, 0, ctypes.byref(p)))


@pytest.mark.parametrize("chunk_mb", [0, 8])
def test_tuned_allocation_tries_candidates_and_releases_the_losers(torch_mod, chunk_mb):
    torch = torch_mod
    vec, texts, ids = _level1(4096, tune=False, engine_options={"obs_chunk_mb": chunk_mb})
    ref = vec.reset().clone()
    eng = vec.engine
    eng.set_option("obs_accept_gbs", 100000)  # out of reach: every candidate is tried unless one is 8 % faster
    torch.cuda.synchronize()
    reserved0 = torch.cuda.memory_reserved()
    st, view, idx, cand = eng.alloc_obs_tuned(vec.puzzle_id, vec.pos, 3)
    torch.cuda.synchronize()
    assert torch.cuda.memory_reserved() == reserved0  # neither the winner nor the losers are torch's
    assert 1 <= len(cand) <= 3 and all(c > 0 for c in cand) and 0 <= idx < 20
    assert torch.equal(view, ref)
    # candidates are screened (a few launches each); the full tuner then runs on the kept one
    assert eng.get_option("tuned_ns") == pytest.approx(min(cand) * 1e6, rel=0.25)
    assert len(cand) >= 2  # 4 096 environments never reach the accept rate: at least two candidates were tuned
    # stepping into the owned buffer
    g = torch.Generator(device=vec.device).manual_seed(11)
    for _ in range(6):
        a = torch.randint(0, 4, (vec.num_envs,), generator=g, device=vec.device, dtype=torch.uint8)
        eng.step_render(vec.puzzle_id, a, vec.pos, vec.steps, vec.reward, vec.dgoals, vec.terminated, vec.truncated, st,
                        vec.flags)
    eng.render(vec.puzzle_id, vec.pos, vec._obs_storage)
    torch.cuda.synchronize()
    assert torch.equal(view, vec.obs)
    envs = [0, 5, 1234, 4095]
    want = _oracle_obs(texts, ids, vec.states(), envs, ref.shape[1] // 3, ref.shape[2] // 3)
    assert np.array_equal(view[envs].cpu().numpy(), want)
    # again and again: candidates come and go (the losers' memory returns to the device, their address ranges stay)
    del st, view
    for _ in range(4):
        reserved0 = torch.cuda.memory_reserved()
        st, view, idx, cand = eng.alloc_obs_tuned(vec.puzzle_id, vec.pos, 3)
        assert torch.cuda.memory_reserved() == reserved0
        assert torch.equal(view, vec.obs)
        del st, view


def test_the_candidate_screen_honours_its_wall_clock_budget(torch_mod):
    """PW_OPT_OBS_TUNE_MS: once the budget is spent no further candidate is allocated -- the best so far is kept and tuned (eight
    ranks of one node screening up to 32 candidates each at the same time must not look like a hang to whoever launched them)."""
    torch = torch_mod
    vec, texts, ids = _level1(4096, tune=False)
    ref = vec.reset().clone()
    eng = vec.engine
    eng.set_option("obs_accept_gbs", 100000)  # out of reach: only the budget (or the 16-candidate rule) ends the screen
    assert eng.get_option("obs_tune_ms") == 10000  # the default
    eng.set_option("obs_tune_ms", 1)              # spent after the first candidate
    st, view, idx, cand = eng.alloc_obs_tuned(vec.puzzle_id, vec.pos, 12)
    assert len(cand) == 1 and torch.equal(view, ref)
    assert 0 <= eng.get_option("obs_screen_ms") < 2000
    del st, view
    eng.set_option("obs_tune_ms", 0)              # back to the default: several candidates again
    st, view, idx, cand = eng.alloc_obs_tuned(vec.puzzle_id, vec.pos, 4)
    assert len(cand) >= 2 and torch.equal(view, ref)


def test_vec_env_binds_its_observation_once(torch_mod):
    torch = torch_mod
    reserved0 = torch.cuda.memory_reserved()
    vec, texts, ids = _level1(4096, tune=True, tune_allocations=2)
    assert vec.tuned_config is not None and 1 <= len(vec.tuned_candidates_ms) <= 2 and vec.tuned_ms > 0
    obs0 = vec.obs
    ptr0 = vec._obs_storage.data_ptr()
    # the library owns the buffer: torch's allocator holds the small state tensors only
    assert torch.cuda.memory_reserved() - reserved0 < vec.num_envs * vec.engine.obs_stride // 2
    out = vec.reset()
    assert out is obs0 and vec._obs_storage.data_ptr() == ptr0
    g = torch.Generator(device=vec.device).manual_seed(3)
    for _ in range(8):
        a = torch.randint(0, 4, (vec.num_envs,), generator=g, device=vec.device, dtype=torch.uint8)
        o = vec.step(a)[0]
        assert o is obs0
    envs = [0, 77, 4095]
    want = _oracle_obs(texts, ids, vec.states(), envs, obs0.shape[1] // 3, obs0.shape[2] // 3)
    assert np.array_equal(obs0[envs].cpu().numpy(), want)
    # a torch-owned buffer tuned in place (tune_allocations = 0) gives the same observations
    alt, _, _ = _level1(4096, tune=True, tune_allocations=0)
    alt.reset()
    alt.set_states(vec.states())
    assert torch.equal(alt.render(), obs0)


def test_vec_env_falls_back_to_a_torch_buffer_when_the_allocator_refuses(torch_mod, monkeypatch):
    """A runtime without the virtual-memory API: the same kernels on a torch-owned buffer, with a warning."""
    torch = torch_mod
    from pushworld_amd import _capi

    def refuse(self, puzzle_id, pos, max_candidates):
        raise RuntimeError("pw_obs_alloc: hipMemCreate: operation not supported")

    monkeypatch.setattr(_capi.Engine, "alloc_obs_tuned", refuse)
    with pytest.warns(RuntimeWarning, match="torch-owned"):
        vec, texts, ids = _level1(4096, tune=True, tune_allocations=2)
    assert not vec.obs_owned_by_library and vec.tuned_config is not None and len(vec.tuned_candidates_ms) == 1
    vec.reset()
    g = torch.Generator(device=vec.device).manual_seed(5)
    for _ in range(5):
        vec.step(torch.randint(0, 4, (vec.num_envs,), generator=g, device=vec.device, dtype=torch.uint8))
    envs = [0, 2048, 4095]
    want = _oracle_obs(texts, ids, vec.states(), envs, vec.obs.shape[1] // 3, vec.obs.shape[2] // 3)
    assert np.array_equal(vec.obs[envs].cpu().numpy(), want)
