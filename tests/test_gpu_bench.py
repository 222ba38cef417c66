"""-m gpu: bench.py as the driver runs it -- the self-spawned multi-rank path, the refusal to
measure fewer ranks than asked, the C4 shard and the shape of the JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "1", "--windows", "2", "--envs-per-gpu", "2048", "--no-extras", "--no-configs"]


def run_bench(*argv, timeout=600, extra_env=None):
    """Runs bench.py; returns (process, [full record]).  What bench.py PRINTS is the compact line (<= 4 KB): it is
    checked here for every run -- one line, short enough for the driver's stdout tail, `roofline` / `cpu_baseline` at the top
    level, the contract fields equal to the full record's -- and the tests then read the full record from the side file."""
    import tempfile

    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.update(extra_env or {})
    with tempfile.TemporaryDirectory() as tmp:
        full_path = os.path.join(tmp, "full.json")
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv, "--full-record", full_path],
                              capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
        fulls = []
        for l in lines:
            assert len(l) < 4096, len(l)
            c = json.loads(l)
            with open(full_path) as f:
                full = json.load(f)
            for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"):
                assert c[key] == full[key], key
            assert c["config"]["workload"] == full["config"]["workload"] and c["full_record"] == full_path
            r = c["roofline"]
            assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
            assert r["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3) and "traffic" in r and "avg_launch_ms" in r
            if "cpu_baseline" in full:
                cb = c["cpu_baseline"]
                assert cb["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-3) and cb["kind"] == "port"
                assert cb["cores"] >= 1 and cb["unit"] == "env-steps/s" and cb["sample"]
            if "configs" in full:
                assert set(c["configs"]) == set(full["configs"]) - {"C3_u8_ppc3"}
            fulls.append(full)
    proc.compact_lines = [json.loads(l) for l in lines]
    return proc, fulls


def test_self_spawned_two_ranks_sum_their_counters():
    """``python bench.py --gpus 2`` (no torchrun) starts two ranks itself.  With one device on the box both ranks
    share it and the counters go over gloo (RCCL refuses two ranks on one GPU); with two or more it is the real
    one-rank-per-GPU RCCL path.  ONE JSON line, n_gpus == 2, counters summed; the cpu_baseline belongs to the N = 1 line."""
    import torch

    shared = [] if torch.cuda.device_count() >= 2 else ["--shared-device"]
    # an N = 1 run of the same workload first: it leaves the reference value of `scaling_efficiency`
    proc1, lines1 = run_bench(*SMALL, "--cpu-seconds", "0.5")
    assert proc1.returncode == 0 and len(lines1) == 1, proc1.stderr[-2000:]
    assert "scaling_efficiency" not in lines1[0] and len(lines1[0]["roofline"]["per_rank_frac"]) == 1
    cb = lines1[0]["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["one_thread"]["cores"] == 1 and cb["cpu_model"]
    assert cb["python_env"]["value"] > 0 and cb["python_env"]["processes"]["cores"] >= 1
    assert proc1.compact_lines[0]["cpu_baseline"]["one_thread_value"] == pytest.approx(cb["one_thread"]["value"], rel=1e-3)
    assert proc1.compact_lines[0]["config"]["ranks_in_probe_all_reduce"] is None
    proc, lines = run_bench("--gpus", "2", *shared, *SMALL, "--cpu-seconds", "1")
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 2 * 2048 and d["config"]["envs_per_gpu"] == 2048
    # value = (sum over ranks of envs x steps) / median window: both ranks' steps are in it
    assert abs(d["value"] - 2 * 2048 * 3 / (d["ms_per_step"] * 3 / 1000.0)) < 1e-6 * d["value"]
    assert len(d["timing"]["per_rank_median_ms_per_step"]) == 2 and len(d["timing"]["window_ms_per_step"]) == 2
    assert d["timing"]["min_ms_per_step"] <= d["ms_per_step"] <= d["timing"]["max_ms_per_step"]
    # with two devices the RCCL branch must have run (init_process_group("nccl") + a probe all_reduce; a failure
    # aborts the run with exit code 3, it never falls back)
    assert ("gloo" if shared else "nccl") in d["config"]["parallelism"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["launches_timed"] == 6 and r["traffic"] is None
    # every rank reports its own dominant-kernel time, roofline fraction and what its allocator found
    assert len(r["per_rank_frac"]) == 2 and all(0 < f < 1 for f in r["per_rank_frac"])
    assert len(r["per_rank_avg_launch_ms"]) == 2 and r["per_rank_avg_launch_ms"][0] == pytest.approx(r["avg_launch_ms"])
    pr = d["config"]["render_launch"]["per_rank"]
    assert len(pr) == 2 and all(set(x) == {"tuned_ms", "allocations_tried", "fast_class", "screen_s", "constructor_s"} for x in pr)
    assert len(d["timing"]["numa_node_per_rank"]) == 2
    se = d["scaling_efficiency"]
    assert se["n1_value"] == pytest.approx(lines1[0]["value"]) and "on this host" in se["n1_source"]
    assert se["value"] == pytest.approx(d["value"] / (2 * lines1[0]["value"]))
    # the counters the step kernels kept on the device, summed over both ranks by the one all_reduce of the job
    c = d["counters"]
    assert c["env_steps"] == 2 * 2048 * 3 * 2 and c["bad_actions"] == 0 and 0 <= c["episodes_solved"] <= c["episodes_ended"]
    assert lines1[0]["counters"]["env_steps"] == 2048 * 3 * 2
    # both statistics: per window the slowest rank (the headline) and every rank's own median rate, summed
    assert d["timing"]["sum_of_per_rank_median_rates"] >= d["value"] * 0.999
    assert se["sum_of_per_rank_rates_over_n_x_n1"] >= se["value"] * 0.999
    assert "cpu_baseline" not in d  # (rank 0 at N = 1 only)
    if not shared:
        assert d["config"]["ranks_in_probe_all_reduce"] == 2


def test_rccl_branch_runs_with_one_rank():
    """The N > 1 plumbing over RCCL on the one device a test box has: BENCH_FORCE_DIST=1 makes a single rank take the
    collective path -- init_process_group("nccl", device_id=...), the probe all_reduce, barriers, the SUM / MAX
    reductions and the per-rank gathers on device tensors -- so that the first 8-GPU run is not the first time that code
    executes."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    proc, lines = run_bench(*SMALL, "--no-cpu-baseline", timeout=600,
                            extra_env={"BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 1 and "counters over nccl" in d["config"]["parallelism"]
    assert d["config"]["ranks_in_probe_all_reduce"] == 1 and proc.compact_lines[0]["config"]["ranks_in_probe_all_reduce"] == 1
    assert d["config"]["global_batch"] == 2048 and abs(d["value"] - 2048 * 3 / (d["ms_per_step"] * 3 / 1000.0)) < 1e-6 * d["value"]
    assert len(d["roofline"]["per_rank_frac"]) == 1 and len(d["timing"]["per_rank_median_ms_per_step"]) == 1


def test_eight_ranks_sharing_the_device_report_eight_rows():
    """The N = 8 line's plumbing on the one GPU a test box has (``--shared-device``: gloo for the reductions): eight
    per-rank rows everywhere the line reports per rank, eight NUMA entries, counters summed over eight ranks."""
    proc, lines = run_bench("--gpus", "8", "--shared-device", "--steps", "2", "--warmup", "1", "--windows", "2", "--envs-per-gpu",
                            "1024", "--no-extras", "--no-configs", "--no-cpu-baseline", timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 * 1024
    assert len(d["roofline"]["per_rank_frac"]) == 8 and len(d["roofline"]["per_rank_avg_launch_ms"]) == 8
    assert len(d["config"]["render_launch"]["per_rank"]) == 8
    assert len(d["timing"]["per_rank_median_ms_per_step"]) == 8 and len(d["timing"]["numa_node_per_rank"]) == 8
    assert d["counters"]["env_steps"] == 8 * 1024 * 2 * 2
    # every rank says what its constructor and, inside it, the allocator's candidate screen cost; the screen stays within its
    # wall-clock budget (PW_OPT_OBS_TUNE_MS) plus one candidate -- eight ranks screening at once on one node must not look like a hang
    rl = d["config"]["render_launch"]
    assert rl["screen_budget_s"] == 10.0 and "fast_class" in rl and rl["constructor_s"] > 0
    for row in rl["per_rank"]:
        assert 0 <= row["screen_s"] <= rl["screen_budget_s"] + 2.0 and row["screen_s"] <= row["constructor_s"] + 0.01, row
    # a stale or missing N = 1 record is refused, not labelled
    se = d["scaling_efficiency"]
    assert se["value"] is None or "on this host" in se["n1_source"]


def test_more_ranks_than_devices_is_refused():
    """--gpus N with fewer than N devices must fail, not report n_gpus 1 (missing #1)."""
    import torch

    n = torch.cuda.device_count() + 1
    proc, lines = run_bench("--gpus", str(n), *SMALL, "--no-cpu-baseline", timeout=120)
    assert proc.returncode != 0 and not lines
    assert "device" in proc.stderr


def test_single_rank_line_has_the_contract_fields():
    proc, lines = run_bench(*SMALL, "--no-cpu-baseline")
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1
    d = lines[0]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "timing"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["dtype"] == "u8" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("C3") and d["roofline"]["kernel"].startswith("pw_render")


def test_extras_of_the_line_run():
    """The non-headline extras of the line (incremental render, state-only rollouts) run and report numbers, not errors."""
    proc, lines = run_bench("--steps", "3", "--warmup", "1", "--windows", "2", "--envs-per-gpu", "2048", "--no-cpu-baseline",
                            "--no-configs")
    assert proc.returncode == 0, proc.stderr[-2000:]
    d = lines[0]
    for key in ("incremental_render", "state_only_rollout"):
        assert key in d and "error" not in d[key], (key, d.get(key))


def test_configs_object_of_the_line():
    """``configs``: the other BASELINE.json configurations on the same clock (here C1, C2 and C5 on `2 Obstacle`; the driver's
    run has all of them): value, kernel, launch time from the library's HIP events, roofline fraction, traffic field."""
    proc, lines = run_bench("--steps", "3", "--warmup", "1", "--windows", "2", "--envs-per-gpu", "2048", "--no-cpu-baseline",
                            "--no-extras", "--configs-only", "C1,C2,C5_2", timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    cfg = lines[0]["configs"]
    assert set(cfg) == {"C1", "C2", "C5_2_obstacle", "C3_u8_ppc3"}, set(cfg)
    for key in ("C1", "C2", "C5_2_obstacle"):
        assert "error" not in cfg[key], cfg[key]
    c1 = cfg["C1"]
    assert c1["gym_step_with_render"]["value"] > 1e3 and c1["get_next_state_batch1"]["value"] > 1e4
    assert c1["render_plan_100_steps"]["frames"] == 101 and c1["render_plan_100_steps"]["value"] < 100.0
    c2 = cfg["C2"]
    assert c2["unit"] == "env-steps/s" and c2["units_per_launch"] == 4096 and c2["algorithmic_bytes_per_unit"] == 38
    assert c2["kernel"] == "pw_step_seg_kernel" and 0 < c2["frac"] < 1 and c2["launches_timed"] == 500  # (every environment bound)
    assert c2["rollout_64_steps_per_launch"]["value"] > c2["value"]
    assert c2["counters"]["env_steps"] > 0 and c2["counters"]["episodes_ended"] > 0
    c5 = cfg["C5_2_obstacle"]
    assert c5["unit"] == "parents/s" and c5["movables"] == 3 and c5["algorithmic_bytes_per_unit"] == 80
    assert c5["states"] * c5["buffer_sets"] * 80 >= 600 << 20 and 0 < c5["frac"] < 1 and "traffic" in c5
    assert cfg["C3_u8_ppc3"]["value"] == lines[0]["value"]


@pytest.mark.parametrize("obs", ["none", "uint8"])
def test_c4_shard_runs(obs):
    """--config c4: the rank's shard of the full mix (all 14 000 Level-0 train puzzles + Levels 1-4, N_pad 32,
    frame 54 x 47), state only and with the uint8 ppc-3 render."""
    proc, lines = run_bench("--config", "c4", "--obs", obs, *SMALL, "--no-cpu-baseline")
    assert proc.returncode == 0, proc.stderr[-2000:]
    d = lines[0]
    assert d["config"]["config"] == "c4" and d["config"]["puzzles"] == 14223 and d["config"]["n_pad"] == 32
    assert d["config"]["frame_cells"] == [54, 47]
    assert d["roofline"]["kernel"] == ("pw_step_group_kernel" if obs == "none" else d["roofline"]["kernel"])
    assert d["value"] > 0
