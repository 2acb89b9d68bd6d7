"""The ONE stdout line the driver parses, as a pure function of the full result dict (tests/test_bench_line.py), and the emitter."""
import json
import os
import sys

from .config import ROOT

LINE_LIMIT = 4096   # bytes: the driver keeps an 8 KiB stdout tail and parses the LAST line (round 4's 25.7 KB line came back `parsed: null`)


def _num(x, sig=6):
    """A float rounded to `sig` significant digits (None / non-numbers pass through)."""
    if isinstance(x, bool) or not isinstance(x, (int, float)):
        return x
    if isinstance(x, int) or x == 0 or x != x or x in (float("inf"), float("-inf")):
        return x
    return float("%.*g" % (sig, x))


def _get(o, *path, default=None):
    for k in path:
        if not isinstance(o, dict) or k not in o or o[k] is None:
            return default
        o = o[k]
    return o


def compact_line(full):
    """The ONE stdout line the driver parses, built from the full result dict: <= LINE_LIMIT bytes, every contract key, `roofline` and
    `cpu_baseline` as flat objects, one number per `also` leg.  Everything else (per-kernel traffic blocks, thread scaling, stage times,
    notes) lives in the full file (`full`: gpurun_out/bench_full.json) and on stderr.  Pure function of its argument: tests/test_bench_line.py
    feeds it canned dicts on the CPU."""
    roof = full.get("roofline") or None
    r = None
    if roof:
        tr = roof.get("traffic") or None
        lanes = roof.get("estimates_per_launch", 1)
        ric = roof.get("rocprof_in_chain") or None
        r = {
            "bound": roof.get("bound"), "kernel": str(roof.get("kernel", ""))[:96],
            "achieved": _num(roof.get("achieved")), "peak": roof.get("peak"), "unit": roof.get("unit"), "frac": _num(roof.get("frac"), 4),
            # the same kernel on the pipe it executes on: split-operand products run on the 16-bit matrix pipe (2.5 PFLOP/s dense)
            "frac_survey_8d": _num(roof.get("frac_survey_8d"), 4),
            "frac_16bit_pipe": _num(_get(roof, "pipe16", "frac"), 4), "frac_f32_mfma": _num(_get(roof, "f32_mfma", "frac"), 4),
            "basis": str(roof.get("basis", ""))[:330] or None,
            "avg_launch_us": _num(roof.get("avg_launch_us"), 5), "lanes": lanes,
            "traffic": (None if not tr else {"bytes_per_launch": _num(tr.get("bytes_per_launch")), "lanes": tr.get("lanes_per_launch", lanes),
                                             "over_algorithmic": _num(tr.get("over_algorithmic"), 3), "GBs": _num(tr.get("GBs"), 5), "frac_of_8TBs": _num(tr.get("frac_of_8TBs"), 4),
                                             "src": str(tr.get("profile", ""))[:64] or None}),
            "rocprof_in_chain": (None if not ric else {"avg_us": _num(ric.get("avg_us"), 5), "lanes": ric.get("lanes"), "src": str(ric.get("source", ""))[:64]}),
        }
        oc = roof.get("other_contraction")
        if isinstance(oc, list):
            oc = oc[0] if oc else None
        if oc:
            r["other"] = {"kernel": str(oc.get("kernel", ""))[:48], "avg_launch_us": _num(oc.get("avg_launch_us"), 5), "frac": _num(oc.get("frac"), 4),
                          "frac_f32_mfma": _num(oc.get("frac_f32_mfma"), 4)}
        if roof.get("draws"):
            r["draws"] = {"avg_launch_us": _num(_get(roof, "draws", "avg_launch_us"), 5), "GBs": _num(_get(roof, "draws", "achieved_GBs"), 4)}
        we = full.get("whole_estimate") or {}
        r["whole_estimate"] = {"hbm_frac_of_8TBs": _num(we.get("hbm_equiv_frac_of_8TBs"), 4), "f32_mfma_TFs": _num(we.get("f32_mfma_TFs"), 4)}
    cb = full.get("cpu_baseline") or None
    c = None
    if cb:
        c = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "threads": cb.get("threads"),
             "kind": cb.get("kind"), "leg": cb.get("leg"), "cpu": str(cb.get("cpu", ""))[:48], "sample": str(cb.get("sample", ""))[:200],
             "one_thread": _num(_get(cb, "one_thread", "estimates_per_s"))}
    also = None
    if full.get("also"):
        also = {}
        for k, v in full["also"].items():
            if not isinstance(v, dict):
                continue
            if "error" in v:
                also[k] = None
            elif "value" in v:
                also[k] = _num(v["value"], 5)
                if k.endswith("_loop") and "us_per_step" in v:
                    also[k + "_us"] = _num(v["us_per_step"], 4)
            elif "us_per_call" in v:
                also[k] = _num(1e6 / v["us_per_call"], 5)
        if isinstance(full["also"].get("ns_f64"), dict) and "frac_f64_mfma" in full["also"]["ns_f64"]:
            also["ns_f64_frac_f64_mfma"] = _num(full["also"]["ns_f64"]["frac_f64_mfma"], 3)
        also["units"] = "estimates/s (c2 ns_dense ns_stl c5 c3 ns_host_boundary ns_f64), steps/s (*_loop, reference_benchmark_grid[_f64]), calls/s (stein), samples/s (ns_objective_1e5)"
    cfg = dict(full.get("config") or {})
    cfg["launch"] = str(cfg.get("launch", ""))[:200]
    cfg["workload"] = str(cfg.get("workload", ""))[:128]
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["value"] = _num(line["value"], 7)
    line["ms_per_step"] = _num(line["ms_per_step"], 6)
    line["config"] = {k: cfg.get(k) for k in ("workload", "d", "n_mc_per_gpu", "n_mc_total", "family", "launch")}
    line["roofline"] = r
    line["cpu_baseline"] = c
    line["elbo_rel_err_vs_cpu_fp64"] = _num(full.get("elbo_rel_err_vs_cpu_fp64"), 3)
    line["grad_rel_l2_vs_cpu_fp64"] = _num(_get(full, "parity_vs_fp64_oracle", "grad_rel_l2"), 3)
    line["steady_state_est_per_s"] = _num(_get(full, "steady_state", "estimates_per_s"), 6)
    line["repeat_ms_per_step"] = [_num(x, 4) for x in (full.get("repeat_ms_per_step") or [])][:5]
    line["also"] = also
    if full.get("dist"):
        d = full["dist"]
        line["dist"] = {"route": d.get("route"), "pipeline": str(d.get("pipeline", ""))[:64] or None,
                        "estimate_sharded_est_per_s": _num(_get(d, "estimate_sharded", "value"), 6),
                        "us_per_estimate": d.get("us_per_estimate"),
                        "p2p_verified": _get(d, "p2p_vs_allreduce", "verified"),
                        "also": ({k: (None if "error" in v else _num(v.get("value"), 5)) for k, v in d["also"].items()} if isinstance(d.get("also"), dict) else None)}
    line["full"] = full.get("full_path")
    s = json.dumps(line, separators=(",", ":"))
    # belt and braces: shed optional blocks, largest first, until the line fits
    for k in ("repeat_ms_per_step", "also", "dist", "steady_state_est_per_s"):
        if len(s) <= LINE_LIMIT:
            break
        line.pop(k, None)
        s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_LIMIT:
        for blk, key in (("roofline", "kernel"), ("cpu_baseline", "sample"), ("config", "launch"), ("config", "workload")):
            if isinstance(line.get(blk), dict) and key in line[blk]:
                line[blk][key] = str(line[blk][key])[:40]
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= LINE_LIMIT, len(s)
    return s


def emit(full, root=None):
    """Full result -> <root>/gpurun_out/bench_full.json (+ stderr), compact line -> stdout (the last thing written there)."""
    root = ROOT if root is None else root
    path = os.path.join(root, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full["full_path"] = os.path.relpath(path, root)
    except OSError:
        full["full_path"] = None
    sys.stderr.write("bench.py full result: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    sys.stdout.write(compact_line(full) + "\n")
    sys.stdout.flush()


