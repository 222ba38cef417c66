"""CPU: the measurement plumbing around bench.py that needs no GPU -- the CPU-baseline legs (tools/cpu_baselines.py: the only
place outside tests/ and smoke() that runs the oracle), the kernel-source hash that guards the PMC records, and the record
lookup of tools/config_suite.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L1 = os.path.join(ROOT, "pushworld_amd", "data", "puzzles", "level1")


def _texts(n=3):
    names = sorted(f for f in os.listdir(L1) if f.endswith(".pwp"))[:n]
    return [open(os.path.join(L1, f)).read() for f in names]


def test_port_rollout_rate_reports_pinned_best_of_three():
    from tools import cpu_baselines as cb

    before = os.sched_getaffinity(0)
    texts = _texts()
    ids = np.arange(96) % len(texts)
    for render in (0, "u8", "f32"):
        out = cb.port_rollout_rate(texts, ids, 50, render, 51, 42, 3, 1, seconds=0.05, samples=3, sample_envs=96)
        assert out["kind"] == "port" and out["unit"] == "env-steps/s" and out["value"] == max(out["samples"]) > 0
        assert len(out["samples"]) == 3 and 1 <= out["cores"] <= cb.hardware_threads()
        assert out["one_thread"]["cores"] == 1 and len(out["one_thread"]["samples"]) == 3
        assert ("float32" in out["sample"]) == (render == "f32") and ("no observation" in out["sample"]) == (render == 0)
    # the calling thread's own affinity mask is put back after every pinned run
    assert os.sched_getaffinity(0) == before
    order = cb.core_first_cpu_order()
    assert sorted(order) == sorted(before) and 1 <= cb.physical_cores() <= len(order)


def test_port_expand_rate_and_python_env_rate():
    from oracle import c_oracle
    from tools import cpu_baselines as cb

    text = open(os.path.join(L1, "2 Obstacle.pwp")).read()
    pz = c_oracle.COraclePuzzle(text, order="cpp")
    st = np.repeat(np.array([[x * 10000 + y for x, y in pz.initial_state]], np.int32), 4096, axis=0)
    out = cb.port_expand_rate(text, st, seconds=0.05)
    assert out["unit"] == "parents/s" and out["value"] == max(out["samples"]) > 0 and len(out["samples"]) == 3
    # in-place outputs give the same answers as fresh ones
    fresh = c_oracle.expand4_batch(pz, st[:64])
    again = c_oracle.expand4_batch(pz, st[:64], out=tuple(np.zeros_like(a) for a in fresh))
    assert all((a == b).all() for a, b in zip(fresh, again))
    py = cb.python_env_rate([text], 50, True, 13, 13, 3, 1, seconds=0.2)
    assert py["value"] > 0 and py["cores"] == 1 and "processes" not in py


def test_pmc_records_are_refused_for_another_kernel_source(tmp_path, monkeypatch):
    from tools import config_suite as cs

    sha = cs.csrc_sha()
    assert len(sha) == 16 and sha == cs.csrc_sha()
    rec = {"configs": {"C4_state": {"kernel_symbol": "k", "units_per_launch": 65536, "hbm_bytes_per_launch": 1.0e7}},
           "csrc_sha16": sha, "source": "test", "git_head": "x"}
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "pmc_kernels_latest.json").write_text(json.dumps(rec))
    monkeypatch.setattr(cs, "ROOT", str(tmp_path))
    t, why = cs.pmc_traffic("C4_state", 65536)
    assert t == 1.0e7 and "same kernel source" in why
    assert cs.pmc_traffic("C4_state", 4096)[0] is None          # another launch size
    assert cs.pmc_traffic("C2_step", 4096)[0] is None           # no record for the configuration
    rec["csrc_sha16"] = "0" * 16
    (prof / "pmc_kernels_latest.json").write_text(json.dumps(rec))
    t, why = cs.pmc_traffic("C4_state", 65536)
    assert t is None and "stale" in why
    # the state-only bytes model of SURVEY 8d
    assert [cs.state_bytes(n) for n in (4, 16, 32)] == [38, 86, 150]


def test_compact_bench_line_fits_the_driver_tail():
    """what bench.py prints is at most 4 KB whatever the full record holds -- here round 4's own 22 KB record
    (which the driver could not parse), the same with eight ranks' per-rank rows and with every optional object bloated."""
    from tools.bench_line import MAX_LINE_BYTES, compact_line, dumps

    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    line = dumps(compact_line(full, "gpurun_out/bench_full.json"))
    assert len(line) <= MAX_LINE_BYTES < 4096
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data"):
        assert d[key] == full[key], key  # the contract's fields keep every digit
    r = d["roofline"]
    assert r["kernel"] == "pw_render_page_kernel" and r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - full["roofline"]["frac"]) < 1e-4 and abs(r["traffic_ratio"] - 1.0045) < 1e-3
    assert r["avg_launch_ms"] > 0 and r["algorithmic_bytes_per_launch"] == full["roofline"]["algorithmic_bytes_per_launch"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 128 and cb["cpu_model"] and cb["one_thread_value"] > 0 and cb["python_env_value"] > 0
    assert d["config"]["workload"] == full["config"]["workload"] and d["config"]["envs_per_gpu"] == 65536
    assert set(d["configs"]) == set(full["configs"]) - {"C3_u8_ppc3"}
    for name, e in d["configs"].items():
        assert e["value"] > 0 and e["unit"], name
    assert d["configs"]["C4_state"]["rollout64"]["value"] > d["configs"]["C4_state"]["value"]
    # eight ranks and bloated optional parts: still inside the limit, the three contract objects still there
    big = json.loads(json.dumps(full))
    big["n_gpus"] = 8
    big["timing"]["per_rank_median_ms_per_step"] = [0.5696312345] * 8
    big["config"]["workload"] = big["config"]["workload"] + " x" * 300
    for e in big["configs"].values():
        e["kernel"] = "pw_some_kernel_with_a_very_long_name<template, arguments, of, all, kinds>(Args)"
        e["hbm_frac"], e["traffic_ratio"] = 0.123456789, 1.23456789
    line8 = dumps(compact_line(big, "gpurun_out/bench_full_n8.json"))
    assert len(line8) <= MAX_LINE_BYTES
    d8 = json.loads(line8)
    assert d8["value"] == full["value"] and "roofline" in d8 and "cpu_baseline" in d8 and d8["config"]["envs_per_gpu"] == 65536
