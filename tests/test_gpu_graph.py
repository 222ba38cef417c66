"""HIP-graph capture of the hot path (torch.cuda.CUDAGraph over the library's launches): the launches go to
torch's current stream and never allocate or synchronise, so a step + render sequence can be captured once
and replayed; results equal the eager path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("incremental", [False, True])
def test_captured_steps_equal_eager(golden, incremental):
    import torch
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    keys = [k for k in golden.keys if k.startswith("bench:level1/")][::4]
    pool = [PushWorldPuzzle(text=golden.text(k)) for k in keys]
    B, K, R = 1024, 4, 12
    ids = np.arange(B) % len(pool)
    kw = dict(puzzle_ids=ids, max_steps=15, pixels_per_cell=3, border_width=1, observation="uint8", autoreset=True,
              incremental=incremental)
    eager, graphed = VecPushWorld(pool, B, **kw), VecPushWorld(pool, B, **kw)
    eager.reset()
    graphed.reset()
    dev = eager.device
    gen = torch.Generator(device=dev).manual_seed(6)
    acts = torch.randint(0, 4, (R, K, B), dtype=torch.uint8, device=dev, generator=gen)
    static_actions = torch.zeros((K, B), dtype=torch.uint8, device=dev)
    # warm-up on a side stream (first calls may allocate scratch), then capture K steps
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        static_actions.copy_(acts[0])
        for k in range(K):
            graphed.step(static_actions[k])
    torch.cuda.current_stream(dev).wait_stream(s)
    for k in range(K):
        eager.step(acts[0, k])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            graphed.step(static_actions[k])
    for r in range(1, R):
        static_actions.copy_(acts[r])
        g.replay()
        for k in range(K):
            eager.step(acts[r, k])
        assert torch.equal(eager.pos, graphed.pos), r
        assert torch.equal(eager.obs, graphed.obs), r
        assert torch.equal(eager.reward, graphed.reward) and torch.equal(eager.terminated, graphed.terminated), r
        assert torch.equal(eager.steps, graphed.steps), r
