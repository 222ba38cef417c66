"""-m gpu: long trajectories.  The other parity tests follow golden sequences of at most a few hundred steps; here
every benchmark puzzle (Levels 1-4) and a Level-0 sample play 3 000 random steps with next-step autoreset and a step
limit, T steps per launch (pw_rollout with histories) -- every reward (float64 bits), terminated and truncated flag
of every step and the final positions against the oracle's trace of the same action tape."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pool,options", [("levels", {}), ("levels", {"step_lds_tables": 1}), ("levels", {"step_wide_groups": 1}),
                                          ("level0", {}), ("level1", {})])
def test_three_thousand_steps_match_the_oracle(pool, options):
    import torch

    from oracle import c_oracle
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    if pool == "levels":
        texts = [open(p).read() for lv in (1, 2, 3, 4) for p in bd.level_paths(lv)]          # N_pad 32
    elif pool == "level1":
        texts = [open(p).read() for p in bd.level_paths(1)]                                  # N_pad 16
    else:
        texts = list(bd.level0_texts(limit=40).values())                                     # N_pad 8, 280 puzzles
    puzzles = [PushWorldPuzzle(text=t) for t in texts]
    oracles = [c_oracle.COraclePuzzle(t, order="python") for t in texts]
    B, T, max_steps = 4 * len(texts), 3000, 150
    ids = np.arange(B) % len(texts)
    acts = np.random.default_rng(11).integers(0, 4, (T, B)).astype(np.uint8)
    vec = VecPushWorld(puzzles, B, puzzle_ids=ids, observation=None, max_steps=max_steps, autoreset=True, device=0,
                       engine_options=options)
    vec.reset()
    pos, rew, term, trunc, steps = c_oracle.rollout_trace(oracles, ids, acts, max_steps, True, vec.num_objects_padded)
    dev = torch.as_tensor(acts).to(vec.device)
    for lo in range(0, T, 500):  # 500 steps per launch, the state carried between the launches
        rh, th, uh = vec.rollout(dev[lo:lo + 500].contiguous(), history=True)
        assert (rh.cpu().numpy().view(np.uint64) == rew[lo:lo + 500].view(np.uint64)).all(), (pool, lo)
        assert (th.cpu().numpy() == term[lo:lo + 500]).all() and (uh.cpu().numpy() == trunc[lo:lo + 500]).all(), (pool, lo)
        assert (vec.pos.cpu().numpy() == pos[lo + 499]).all() and (vec.steps.cpu().numpy() == steps[lo + 499]).all(), (pool, lo)
    assert term.sum() > 0 and trunc.sum() > 0  # episodes did end both ways
