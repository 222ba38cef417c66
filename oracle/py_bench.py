"""CPU-baseline driver for bench.py (test infrastructure, like everything under oracle/): the pure-Python
restatement of the reference environment (oracle/pw_oracle.py) stepped serially for a fixed wall time, in this
process (``run``) or in P independent worker processes (``run_many``).  Workers are started with
the "spawn" method: the caller has a live HIP context that must not be forked."""
import multiprocessing as mp
import time

import numpy as np


def run(job):
    """job: texts, max_steps, render, pad_h, pad_w, ppc, bw, seconds, seed, obs_dtype ("u8" default / "f32").  Returns steps, seconds and the time
    spent building the collision tables (puzzle.py:259-311 restated; not part of the rate)."""
    from oracle import pw_oracle

    rng = np.random.default_rng(job.get("seed", 0))
    t0 = time.perf_counter()
    envs = [pw_oracle.OracleEnv(pw_oracle.OraclePuzzle(t), job["max_steps"]) for t in job["texts"]]
    build = time.perf_counter() - t0
    for e in envs:
        e.reset()
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < job["seconds"]:
        for e in envs:
            state, _, term, trunc = e.step(int(rng.integers(0, 4)))
            if job["render"]:
                if job.get("obs_dtype") == "f32":  # what gym_env.py:188-226 returns: uint8 -> float32 / 255, zero padded
                    e.puzzle.observation(state, job["pad_h"], job["pad_w"], job["ppc"], job["bw"])
                else:
                    e.puzzle.observation_u8(state, job["pad_h"], job["pad_w"], job["ppc"], job["bw"])
            if term or trunc:
                e.reset()
            steps += 1
    return {"steps": steps, "seconds": time.perf_counter() - t0, "build_seconds": build}


def _worker(args):
    job, seed = args
    return run(dict(job, seed=seed))


def run_many(job, processes):
    """``processes`` workers, each running ``run(job)`` with its own seed; the rate is the sum of the workers'
    own rates (start-up and table construction excluded, as in ``run``)."""
    ctx = mp.get_context("spawn")
    with ctx.Pool(processes) as pool:
        res = pool.map(_worker, [(job, 1000 + i) for i in range(processes)], chunksize=1)
    return {"processes": processes, "steps_per_s": float(sum(r["steps"] / r["seconds"] for r in res))}
