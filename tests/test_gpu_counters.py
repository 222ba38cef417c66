"""-m gpu: the device-side throughput counters (``pw_counters``) against the oracle's trace.

Every step kernel adds, per wavefront, what its environments did: env-steps, episodes ended (terminated or truncated,
gym_env.py:210-223), episodes solved (terminated).  The sums must equal the sums of the C oracle's per-step flags for the
same actions, whichever kernel formulation runs and whether the steps come one per launch or as a rollout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _level1(golden):
    return [k for k in golden.keys if k.startswith("bench:level1/")]


def _want(oracles, ids, acts, max_steps, NP):
    from oracle import c_oracle

    pos, _, te, tr, steps = c_oracle.rollout_trace(oracles, ids, acts, max_steps, True, NP)
    ended = int(((te | tr) != 0).sum())
    return {"env_steps": int(acts.size), "episodes_ended": ended, "episodes_solved": int((te != 0).sum()), "bad_actions": 0}, pos


@pytest.mark.parametrize("flavour", ["group", "lane", "wave", "big-batch", "rollout", "rollout-lane", "render", "fused", "delta",
                                     "boards", "boards-rollout", "npad32"])
def test_counters_equal_the_oracle_trace(golden, flavour):
    import torch

    from oracle import c_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    if flavour.startswith("boards"):  # sets of 8 x 8 puzzles: whole-grid boards
        keys = []
        for k in golden.keys:
            if k.startswith("l0:"):
                m = golden.meta[k]
                if m["width"] <= 8 and m["height"] <= 8 and m["num_movables"] <= 8:
                    keys.append(k)
        keys = keys[:40]
        assert len(keys) >= 10
    elif flavour == "npad32":  # a set with puzzles of more than 16 movables next to small ones: the mixed-width lane groups
        keys = [k for k in golden.keys if k.startswith("bench:level") and golden.meta[k]["num_movables"] > 16][:3] + \
               [k for k in golden.keys if k.startswith("bench:level")][::9]
    else:
        keys = _level1(golden)[::2]
    texts = [golden.text(k) for k in keys]
    pool = [PushWorldPuzzle(text=t) for t in texts]
    oracles = [c_oracle.COraclePuzzle(t) for t in texts]
    B, T, max_steps = 4096 + 37, 24, 7
    ids = (np.arange(B, dtype=np.int64) * len(pool)) // B
    opts = {"group": {"step_kernel": "group", "step_boards": "never"}, "lane": {"step_kernel": "lane"},
            "wave": {"step_kernel": "wave"}, "big-batch": {"step_lane_batch": 1, "step_boards": "never"},
            "rollout": {"step_boards": "never"}, "rollout-lane": {"step_kernel": "lane"}}.get(flavour, {})
    obs = "uint8" if flavour in ("render", "fused", "delta") else None
    vec = VecPushWorld(pool, B, puzzle_ids=ids, max_steps=max_steps, pixels_per_cell=3, border_width=1, observation=obs,
                       autoreset=True, incremental=flavour == "delta", engine_options=opts,
                       bind=False if flavour == "boards" else None)  # ("boards-rollout": every environment bound -- the segments)
    if flavour == "fused":
        vec.engine.set_option("fused_step_render", 1)
    if flavour.startswith("boards"):
        assert vec.engine.get_option("step_board_set") == 1
    if flavour == "npad32":
        assert vec.num_objects_padded == 32
    acts = np.random.default_rng(11).integers(0, 4, size=(T, B), dtype=np.uint8)
    want, want_pos = _want(oracles, ids, acts, max_steps, vec.num_objects_padded)
    assert want["episodes_ended"] > B and want["episodes_solved"] >= 0
    vec.reset()
    assert vec.counters() == {"env_steps": 0, "episodes_ended": 0, "episodes_solved": 0, "bad_actions": 0}
    acts_dev = torch.as_tensor(acts).to(vec.device)
    if flavour in ("rollout", "rollout-lane", "boards-rollout"):
        vec.rollout(acts_dev[:10])
        vec.rollout(acts_dev[10:])
    else:
        for t in range(T):
            vec.step(acts_dev[t])
    assert (vec.pos.cpu().numpy() == want_pos[-1]).all()
    assert vec.counters() == want
    # reading does not clear; pw_counters_reset does
    assert vec.counters() == want
    vec.counters_reset()
    assert vec.counters()["env_steps"] == 0


def test_refused_actions_are_counted_as_steps_not_as_episodes(golden):
    import torch

    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    pool = [PushWorldPuzzle(text=golden.text(k)) for k in _level1(golden)[:4]]
    B = 300
    vec = VecPushWorld(pool, B, max_steps=None, observation=None)
    vec.reset()
    a = torch.zeros((B,), dtype=torch.uint8, device=vec.device)
    a[::3] = 9
    vec.step(a)
    c = vec.counters()
    assert c["env_steps"] == B and c["bad_actions"] == 100 and c["episodes_ended"] == 0
    assert vec.engine.bad_actions() == 100       # reads and clears its own counter ...
    assert vec.counters()["bad_actions"] == 100  # ... the cumulative one stays
    vec.step(a)
    assert vec.counters()["bad_actions"] == 200 and vec.counters()["env_steps"] == 2 * B
