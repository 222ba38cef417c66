"""The ONE stdout line of bench.py: at most ``MAX_LINE_BYTES`` bytes.

bench.py builds the full record (every sample, window list, tuner candidate, rocm-smi field, sample description) and writes it
to ``gpurun_out/bench_full.json``; what it PRINTS is ``compact_line(full)``: the contract's top-level fields, ``config``,
``roofline`` and ``cpu_baseline`` as top-level objects, and one short object per other BASELINE.json configuration.  The
driver keeps only the last few KB of stdout -- a line longer than that arrives without its head and parses as nothing
(BENCH_r04.json: ``parsed: null`` for a 22 KB line)."""
import json


def dumps(obj):
    """No spaces after separators: a sixth of the line."""
    return json.dumps(obj, separators=(",", ":"))


MAX_LINE_BYTES = 3900  # (the driver keeps ~8.6 KB of stdout; the judge asked for <= 4 KB: some margin under 4 096)

TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
CONFIG_KEYS = ("workload", "config", "envs_per_gpu", "global_batch", "puzzles", "frame_cells", "pixels_per_cell", "border_width",
               "observation", "obs_shape", "max_steps", "autoreset", "n_pad", "parallelism", "ranks_in_probe_all_reduce",
               "algorithmic_bytes_per_env_step")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "hbm_frac", "traffic", "traffic_ratio", "traffic_from",
                 "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_timed", "timer")
SUB_KEYS = ("value", "unit", "kernel", "avg_launch_ms", "frac", "hbm_frac", "traffic_ratio")


def sig(x, digits=5):
    """Floats to ``digits`` significant digits (the full record keeps every digit)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{digits}g}")


def _round(obj, digits=5):
    if isinstance(obj, dict):
        return {k: _round(v, digits) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_round(v, digits) for v in obj]
    return sig(obj, digits)


def compact_roofline(r):
    out = {k: r[k] for k in ROOFLINE_KEYS if k in r}
    if out.get("traffic") and out.get("algorithmic_bytes_per_launch") and "traffic_ratio" not in out:
        out["traffic_ratio"] = out["traffic"] / out["algorithmic_bytes_per_launch"]
    if isinstance(out.get("timer"), str):
        out["timer"] = out["timer"][:60]
    return out


def compact_cpu(c):
    if not isinstance(c, dict) or "error" in c:
        return c
    out = {k: c[k] for k in ("value", "unit", "cores", "kind", "cpu_model", "host_cpus") if k in c}
    if isinstance(c.get("sample"), str):
        out["sample"] = c["sample"][:110]
    if isinstance(c.get("one_thread"), dict):
        out["one_thread_value"] = c["one_thread"].get("value")
    py = c.get("python_env")
    if isinstance(py, dict):
        out["python_env_value"] = py.get("value")
        if isinstance(py.get("processes"), dict):
            out["python_env_processes"] = {"value": py["processes"].get("value"), "cores": py["processes"].get("cores")}
    return out


def compact_sub(e):
    """One non-headline configuration: value, unit, kernel, launch time, model / measured-traffic fractions, CPU value."""
    if not isinstance(e, dict):
        return e
    if "error" in e:
        return {"error": str(e["error"])[:120]}
    out = {k: e[k] for k in SUB_KEYS if k in e and e[k] is not None}
    if isinstance(out.get("kernel"), str):
        out["kernel"] = out["kernel"].split(" ")[0][:40]
    cb = e.get("cpu_baseline")
    if isinstance(cb, dict) and "value" in cb:
        out["cpu_value"] = cb["value"]
        out["cpu_cores"] = cb.get("cores")
    ro = e.get("rollout_64_steps_per_launch")
    if isinstance(ro, dict) and "value" in ro:
        out["rollout64"] = {k: ro[k] for k in ("value", "avg_launch_ms", "frac", "hbm_frac", "traffic_ratio") if ro.get(k) is not None}
    mb = e.get("mailbox_step")
    if isinstance(mb, dict) and "value" in mb:
        out["mailbox"] = {k: mb[k] for k in ("value", "us_per_step") if mb.get(k) is not None}
        if mb.get("us_per_step_8_in_flight") is not None:  # (the host posts ahead of its waits: the resident kernel's own rate)
            out["mailbox"]["us8"] = round(float(mb["us_per_step_8_in_flight"]), 3)
    return out


def compact_line(full, full_path=None):
    """The dict bench.py prints.  Shrinks optional parts until ``json.dumps`` fits ``MAX_LINE_BYTES``."""
    out = {k: full.get(k) for k in TOP_KEYS}
    cfg = full.get("config", {})
    out["config"] = {k: cfg[k] for k in CONFIG_KEYS if k in cfg}
    rl = cfg.get("render_launch")
    if isinstance(rl, dict):
        # (which of the two buffer classes the allocator found decides 0.86 vs 0.875 of peak: a slow-class box must not read as a
        # regression; constructor_s / screen_s: what the screen of up to 32 candidates cost)
        out["config"]["render_launch"] = {k: rl[k] for k in ("tuned_index", "tuned_ms", "allocations_tried", "fast_class", "constructor_s",
                                                             "screen_s") if k in rl}
    if "roofline" in full:
        out["roofline"] = compact_roofline(full["roofline"])
    if "cpu_baseline" in full:
        out["cpu_baseline"] = compact_cpu(full["cpu_baseline"])
    t = full.get("timing", {})
    out["timing"] = {k: t[k] for k in ("windows", "steps_per_window", "statistic", "min_ms_per_step", "max_ms_per_step",
                                       "per_rank_median_ms_per_step") if k in t}
    if "counters" in full:
        out["counters"] = full["counters"]
    dev = full.get("device", {})
    if dev.get("name"):
        out["device"] = dev["name"]
    if isinstance(full.get("scaling_efficiency"), dict):
        out["scaling_efficiency"] = {k: full["scaling_efficiency"].get(k) for k in ("value", "n1_value") if k in full["scaling_efficiency"]}
    for k in ("incremental_render", "state_only_rollout"):
        if isinstance(full.get(k), dict) and "env_steps_per_s" in full[k]:
            out[k] = {"env_steps_per_s": full[k]["env_steps_per_s"]}
    if isinstance(full.get("configs"), dict):
        # (the headline's own entry repeats the top-level fields: left to the full record)
        out["configs"] = {name: compact_sub(e) for name, e in full["configs"].items() if name != "C3_u8_ppc3"}
    if full_path:
        out["full_record"] = full_path
    top = {k: out[k] for k in TOP_KEYS}  # the contract's own fields keep every digit
    out = _round(out)
    out.update(top)
    # shrink until it fits: first the nice-to-haves, then digits, at last whole optional objects
    for drop in (("incremental_render", "state_only_rollout"), ("device", "scaling_efficiency"), ("timing",), ("counters",)):
        if len(dumps(out)) <= MAX_LINE_BYTES:
            break
        if drop == ("timing",) and "per_rank_median_ms_per_step" in out.get("timing", {}):
            out["timing"].pop("per_rank_median_ms_per_step")
            continue
        for k in drop:
            out.pop(k, None)
    if len(dumps(out)) > MAX_LINE_BYTES and "configs" in out:
        for e in out["configs"].values():
            if isinstance(e, dict):
                for k in ("rollout64", "mailbox", "cpu_cores", "avg_launch_ms"):
                    e.pop(k, None)
    if len(dumps(out)) > MAX_LINE_BYTES:
        out = _round(out, 4)
        out.update(top)
    if len(dumps(out)) > MAX_LINE_BYTES and "configs" in out:
        out["configs"] = {name: ({"value": e.get("value"), "frac": e.get("frac")} if isinstance(e, dict) else e)
                          for name, e in out["configs"].items()}
    if len(dumps(out)) > MAX_LINE_BYTES:
        out.pop("configs", None)
        out["config"]["workload"] = str(out["config"].get("workload", ""))[:120]
    return out
