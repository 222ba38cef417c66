#!/usr/bin/env python3
"""Library-owned observation buffers (pw_obs_alloc / pw_obs_alloc_tuned) on the GPU box.

    obs_alloc_xp.py wrap                 torch adoption of the buffer, free / allocate again cycles
    obs_alloc_xp.py one  [c3|c4] [k] [chunk_mb]   ONE fresh process: the product flow (VecPushWorld), prints a JSON line
    obs_alloc_xp.py many [c3|c4] [n] [k] [chunk_mb]  n fresh processes of `one`
    obs_alloc_xp.py classes [c3|c4] [chunk_mb] [n]   n buffers alive at once in one process: tuned ms of each
"""
import json
import os
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def workload(config, k, chunk_mb):
    import bench
    from pushworld_amd import vec_env

    args = SimpleNamespace(envs_per_gpu=65536, obs="uint8", config=config, max_steps=200, bw=1, ppc=3, tune_allocations=k)
    if chunk_mb:  # engine option before the allocation: through VecPushWorld's engine_options
        orig = vec_env.VecPushWorld.__init__

        def init(self, *a, **kw):
            kw.setdefault("engine_options", {})["obs_chunk_mb"] = chunk_mb
            orig(self, *a, **kw)

        vec_env.VecPushWorld.__init__ = init
    t0 = time.perf_counter()
    wl = bench.build_workload(args, 0, 1, 0)
    return wl, time.perf_counter() - t0


def one(config, k, chunk_mb):
    wl, dt = workload(config, k, chunk_mb)
    vec = wl["vec"]
    eng = vec.engine
    vec.reset()
    B = vec.num_envs
    acts = torch.randint(0, 4, (64, B), device=vec.device, dtype=torch.uint8)
    for t in range(5):
        vec.step(acts[t])
    eng.profile_render(50)
    for t in range(50):
        vec.step(acts[t % 64])
    ms = np.array(eng.profile_read())
    eng.profile_render(0)
    print(json.dumps({"config": config, "tuned_ms": round(vec.tuned_ms, 4), "loop_ms": round(float(np.median(ms)), 4),
                      "tried": len(vec.tuned_candidates_ms), "cand_ms": [round(x, 4) for x in vec.tuned_candidates_ms],
                      "idx": vec.tuned_config, "load_all": eng.get_option("page_load_all"),
                      "cfg": [eng.get_option(o) for o in ("page_order", "page_run_log2", "page_lds_pad_kb")],
                      "torch_reserved_mb": torch.cuda.memory_reserved() >> 20, "buf_mb": (B * eng.obs_stride) >> 20,
                      "ctor_s": round(dt, 2)}), flush=True)


def classes(config, chunk_mb, n):
    wl, _ = workload(config, 0, chunk_mb)  # plain torch buffer, not tuned
    vec = wl["vec"]
    eng = vec.engine
    vec.reset()
    rows = []
    keep = []
    for i in range(n):
        st, view, idx, cand = eng.alloc_obs_tuned(vec.puzzle_id, vec.pos, 1)
        keep.append((st, view))
        rows.append((round(cand[0], 4), idx))
    print(config, "chunk_mb", chunk_mb, "tuned ms, index:", rows, flush=True)


def soak(n):
    """n constructions / destructions of the C3 environment in ONE process: candidates come and go, the free device memory
    (hipMemGetInfo through the runtime torch loaded) must return to where it was, nothing may fault."""
    import ctypes
    import gc

    hip = ctypes.CDLL("libamdhip64.so")

    def free_mb():
        fr, tot = ctypes.c_size_t(), ctypes.c_size_t()
        hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot))
        return fr.value >> 20

    base = None
    for i in range(n):
        wl, dt = workload("c3", None, 0)
        vec = wl["vec"]
        vec.reset()
        acts = torch.randint(0, 4, (4, vec.num_envs), device=vec.device, dtype=torch.uint8)
        for t in range(4):
            vec.step(acts[t])
        torch.cuda.synchronize()
        row = (round(vec.tuned_ms, 4), len(vec.tuned_candidates_ms), round(dt, 2), free_mb())
        del vec, wl, acts
        gc.collect()
        torch.cuda.synchronize()
        after = free_mb()
        if base is None:
            base = after
        print("cycle %2d: tuned %.4f ms, %d candidates, ctor %.2f s, free while alive %d MB, after %d MB (first cycle: %d)"
              % (i, row[0], row[1], row[2], row[3], after, base), flush=True)
    print("soak ok" if abs(after - base) < 512 else "LEAK: %d MB" % (base - after))


def wrap():
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld
    import bench

    paths = bench.level1_paths()
    B = 8192
    ids = (np.arange(B, dtype=np.int64) * len(paths)) // B
    vec = VecPushWorld([PushWorldPuzzle(p) for p in paths], B, puzzle_ids=ids, max_steps=200, pixels_per_cell=3,
                       border_width=1, observation="uint8", tune=False)
    ref = vec.reset().clone()
    eng = vec.engine
    print("torch reserved before: %d MB" % (torch.cuda.memory_reserved() >> 20))
    for cycle in range(6):
        st, view = eng.alloc_obs_owned(B)
        assert int(st.abs().sum()) == 0
        eng.render(vec.puzzle_id, vec.pos, st)
        torch.cuda.synchronize()
        assert torch.equal(view, ref), "mismatch in cycle %d" % cycle
        host = view[:4].cpu()
        assert torch.equal(host, ref[:4].cpu())
        free0 = torch.cuda.mem_get_info()[0]
        ptr = st.data_ptr()
        del st, view, host
        free1 = torch.cuda.mem_get_info()[0]
        junk = torch.empty((64 << 20,), dtype=torch.uint8, device=vec.device)  # something else takes memory in between
        print("cycle %d ptr %#x freed %d MB" % (cycle, ptr, (free1 - free0) >> 20), flush=True)
        del junk
    # tuned allocation, several candidates forced (accept threshold out of reach)
    eng.set_option("obs_accept_gbs", 100000)
    for cycle in range(3):
        st, view, idx, cand = eng.alloc_obs_tuned(vec.puzzle_id, vec.pos, 3)
        torch.cuda.synchronize()
        assert torch.equal(view, ref)
        print("tuned cycle %d: idx %d candidates %s" % (cycle, idx, [round(c, 4) for c in cand]), flush=True)
        del st, view
    print("torch reserved after: %d MB; wrap ok" % (torch.cuda.memory_reserved() >> 20))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "wrap":
        wrap()
    elif mode == "one":
        cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
        k = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
        one(cfg, k, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    elif mode == "many":
        cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
        rest = sys.argv[4:]
        for i in range(n):
            try:
                subprocess.run([sys.executable, os.path.abspath(__file__), "one", cfg] + rest, timeout=200)
            except subprocess.TimeoutExpired:
                print("TIMEOUT", flush=True)
    elif mode == "soak":
        soak(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    elif mode == "classes":
        cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
        classes(cfg, int(sys.argv[3]) if len(sys.argv) > 3 else 0, int(sys.argv[4]) if len(sys.argv) > 4 else 8)
