"""CPU baselines of bench.py (the ``cpu_baseline`` leg: the only place besides tests/ and smoke() that runs anything under
``oracle/``): the oracle's C port on this host's cores, on a bounded sample of the same workload.

Stability: every OpenMP thread pins itself to its own CPU (one hardware thread of every physical core
first, then the SMT siblings: ``core_first_cpu_order``; OMP_PROC_BIND in the environment would also bind the thread that
launches the GPU kernels), every figure is the BEST of three samples, and all three values are reported."""
import ctypes
import os
import time

import numpy as np


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def hardware_threads():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def core_first_cpu_order():
    """The CPUs this process may use, one hardware thread of every physical core first (in core order), then their SMT
    siblings: thread t of a run pinned to entry t lands on its own core as long as there are cores left."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    seen, first, rest = set(), [], []
    for cpu in allowed:
        key = None
        try:
            with open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list") as f:
                key = f.read().strip()
        except OSError:
            key = str(cpu)
        if key in seen:
            rest.append(cpu)
        else:
            seen.add(key)
            first.append(cpu)
    return first + rest


class pinned_threads:
    """Context: the oracle's OpenMP threads pin themselves (``or_set_thread_cpus``); the calling thread's own mask is put back."""

    def __enter__(self):
        from oracle import c_oracle

        self.c_oracle = c_oracle
        try:
            self.saved = os.sched_getaffinity(0)
        except AttributeError:
            self.saved = None
        self.order = core_first_cpu_order()
        c_oracle.set_thread_cpus(self.order)
        return self

    def __exit__(self, *exc):
        self.c_oracle.set_thread_cpus([])
        if self.saved is not None:
            try:
                os.sched_setaffinity(0, self.saved)
            except OSError:
                pass
        return False


def physical_cores():
    order = core_first_cpu_order()
    seen = set()
    for cpu in order:
        try:
            with open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list") as f:
                seen.add(f.read().strip())
        except OSError:
            seen.add(str(cpu))
    return max(1, len(seen))


def set_omp_threads(n):
    """torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks; loops without a num_threads clause (the expansion
    checker) take their thread count from here."""
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def port_rollout_rate(texts, ids_full, max_steps, render, pad_h, pad_w, ppc, bw, seconds=3.0, samples=3, sample_envs=4096,
                      with_one_thread=True):
    """env-steps/s of the C port (``or_rollout``: OpenMP over environments, next-step autoreset) on ``sample_envs``
    environments of the workload's puzzle mix.  ``render``: 0 state only, 1 / "u8" + padded uint8 observation, 2 / "f32"
    + float32 observation.  Returns the ``cpu_baseline`` object."""
    from oracle import c_oracle

    B = min(sample_envs, len(ids_full))
    stride = max(1, len(ids_full) // B)
    ids_sample = np.asarray(ids_full[::stride][:B], dtype=np.int64)
    used = np.unique(ids_sample)  # only the puzzles the sample touches are compiled
    remap = {int(p): i for i, p in enumerate(used)}
    puzzles = [c_oracle.COraclePuzzle(texts[int(p)]) for p in used]
    ids = np.asarray([remap[int(p)] for p in ids_sample], dtype=np.int32)
    B = len(ids)
    rng = np.random.default_rng(12345)

    render = {"u8": 1, "f32": 2}.get(render, render) if isinstance(render, str) else int(render)

    def timed(T, threads):
        acts = rng.integers(0, 4, size=(T, B), dtype=np.uint8)
        with pinned_threads():
            t0 = time.perf_counter()
            _, used_threads = c_oracle.rollout(puzzles, ids, acts, max_steps, render, pad_h, pad_w, ppc, bw, threads=threads)
            dt = time.perf_counter() - t0
        return B * T / dt, used_threads, dt

    def calibrated(threads):
        """a rate from a run long enough (>= 0.25 s) that thread start-up does not dominate"""
        T = 1
        while True:
            rate, used_threads, dt = timed(T, threads)
            if dt >= 0.25 or T >= 16384:
                return rate
            T = max(T + 1, int(T * min(8.0, 0.3 / max(dt, 1e-4))))

    hw = hardware_threads()
    # all hardware threads this process may use, or one per physical core on an SMT-2 host: whichever is faster here
    best_threads, best_rate = 1, 0.0
    for cand in sorted({hw, min(hw, physical_cores())}):
        rate = calibrated(cand)
        if rate > best_rate:
            best_threads, best_rate = cand, rate

    def best_of(threads, rate_guess, secs):
        vals, T_used = [], 0
        for _ in range(samples):
            T2 = int(max(2, min(65536, secs * rate_guess / B)))
            rate, used_threads, dt = timed(T2, threads)
            vals.append(rate)
            T_used = T2
        return max(vals), vals, T_used

    rate, vals, T2 = best_of(best_threads, best_rate, seconds)
    what = {0: "step only (no observation)", 1: f"step + padded uint8 render ppc={ppc}",
            2: f"step + padded float32 render ppc={ppc}"}[render]
    out = {
        "value": rate, "unit": "env-steps/s", "cores": best_threads, "kind": "port",
        "sample": f"{B} envs (same puzzle mix) x {T2} steps, {what}, OpenMP over envs, every thread pinned to its own CPU "
                  f"(one per physical core first), best of {samples} samples of ~{seconds:.1f} s",
        "samples": vals,
    }
    if with_one_thread:
        r1 = calibrated(1)
        rate1, vals1, T1 = best_of(1, r1, max(0.5, seconds * 0.4))
        out["one_thread"] = {"value": rate1, "cores": 1, "sample": f"{B} envs x {T1} steps, best of {samples}", "samples": vals1}
    return out


def port_expand_rate(text, states, seconds=0.5, samples=3):
    """parents/s of the C port's 4-action expansion (``or_expand4_batch``, OpenMP over states) on a sample of the same
    frontier, C++ object order: one thread per physical core or all hardware threads, whichever is faster on this host
    (all 256 hardware threads of the bench boxes are 10x SLOWER than 128 here), every thread pinned, best of three samples."""
    from oracle import c_oracle

    pz = c_oracle.COraclePuzzle(text, order="cpp")
    st = np.ascontiguousarray(states[: min(len(states), 1 << 20)])
    best = None
    candidates = sorted({min(hardware_threads(), physical_cores()), hardware_threads()})  # (before the calling thread is pinned)
    with pinned_threads():
        out = c_oracle.expand4_batch(pz, st)  # (first pass: the output pages are touched here, not in the timed ones)
        for threads in candidates:
            set_omp_threads(threads)
            c_oracle.expand4_batch(pz, st, out)
            vals, reps = [], 0
            for _ in range(samples):  # every sample is bounded by the clock, not by a pass count guessed from one pass
                t0 = time.perf_counter()
                n = 0
                while True:
                    c_oracle.expand4_batch(pz, st, out)
                    n += 1
                    dt = time.perf_counter() - t0
                    if dt >= seconds or n >= 512:
                        break
                vals.append(n * len(st) / dt)
                reps = max(reps, n)
            if best is None or max(vals) > best[0]:
                best = (max(vals), vals, threads, reps)
    rate, vals, threads, reps = best
    return {"value": rate, "unit": "parents/s", "cores": threads, "kind": "port",
            "sample": f"{len(st)} states of the same frontier x up to {reps} passes per sample (~{seconds:.1f} s), or_expand4_batch (OpenMP over "
                      f"states, pinned threads), best of {samples}",
            "samples": vals}


def python_env_rate(texts, max_steps, render, pad_h, pad_w, ppc, bw, seconds=2.0, processes=0, obs_dtype="u8"):
    """The pure-Python restatement of the reference environment (oracle/pw_oracle.py: hash-set collision tables, per-cell
    painter, /255 + np.pad -- the closest thing to the reference's own CPU Python env that can travel to the GPU box), one
    process; with ``processes`` > 0 also that many independent workers."""
    from oracle import py_bench

    job = dict(texts=list(texts), max_steps=max_steps, render=bool(render), pad_h=pad_h, pad_w=pad_w, ppc=ppc, bw=bw,
               seconds=seconds, obs_dtype=obs_dtype)
    one = py_bench.run(dict(job, seed=777))
    what = (f"step + padded {'float32 (/255)' if obs_dtype == 'f32' else 'uint8'} render ppc={ppc}") if render else "step only"
    out = {"value": one["steps"] / one["seconds"], "unit": "env-steps/s", "cores": 1, "kind": "port (pure Python)",
           "sample": f"{len(texts)} puzzle(s) x {one['steps'] // max(1, len(texts))} steps, {what}, {one['seconds']:.1f} s; "
                     f"collision-table construction took {one['build_seconds']:.1f} s (not included)"}
    if processes > 0:
        many = py_bench.run_many(job, processes)
        out["processes"] = {"value": many["steps_per_s"], "cores": many["processes"],
                            "sample": f"{many['processes']} worker processes x {seconds:.0f} s, same job each"}
    return out
