"""bench.py prints ONE stdout line the driver parses from an 8 KiB tail (round 4: a 25.7 KB line came back `parsed: null`).
The line is a pure function of the full result dict (bench.compact_line): canned dicts here, no GPU."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def _round4_line():
    """The last full line of round 4 as the driver saw it (25.7 KB): the worst case on record."""
    return json.load(open(os.path.join(ROOT, "profiles", "r04_e_bench_driver_protocol.json")))


def _check(line_s, full):
    assert len(line_s.encode()) <= bench.LINE_LIMIT <= 4096
    assert "\n" not in line_s
    line = json.loads(line_s)
    assert json.loads(json.dumps(line)) == line
    for k in CONTRACT:
        assert k in line, k
    assert line["metric"] == full["metric"] and line["n_gpus"] == full["n_gpus"] and line["steps"] == full["steps"]
    assert line["value"] == pytest.approx(full["value"], rel=1e-6)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert set(line["config"]) >= {"workload", "d", "n_mc_per_gpu", "family", "launch"} and "model" not in line["config"]
    assert len(line["config"]["launch"]) <= 200
    return line


def test_round4_full_line_compacts_below_4k():
    full = _round4_line()
    assert len(json.dumps(full)) > 20000
    line = _check(bench.compact_line(full), full)
    r, c = line["roofline"], line["cpu_baseline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "rocprof_in_chain"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s", "G wave-instructions/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")
    # one number per `also` leg
    for k in ("c2", "ns_dense", "ns_stl", "c5", "c3", "ns_adam_loop"):
        assert isinstance(line["also"][k], (int, float)), k
    assert line["elbo_rel_err_vs_cpu_fp64"] == pytest.approx(full["elbo_rel_err_vs_cpu_fp64"], rel=1e-2)


def test_minimal_and_multi_gpu_lines():
    base = dict(metric="ELBO-grad-estimates/sec", value=123456.789, unit="estimates/s", n_gpus=8, steps=20, warmup=5, ms_per_step=0.0081,
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload="w" * 500, d=1024, n_mc_per_gpu=256, n_mc_total=2048, family="fullrank", launch="x" * 1000, seed="0x1"),
                roofline=None, cpu_baseline=None, also=None,
                dist=dict(route="p2p", pipeline="y" * 300, estimate_sharded=dict(value=1.0e6), us_per_estimate=dict(partials=1.0, exchange=2.0),
                          per_rank=[dict(rank=r, note="z" * 400) for r in range(8)]))
    line = _check(bench.compact_line(base), base)
    assert line["roofline"] is None and line["cpu_baseline"] is None
    assert line["dist"]["route"] == "p2p" and line["dist"]["estimate_sharded_est_per_s"] == 1.0e6
    # absurdly long strings everywhere still fit
    fat = dict(base, roofline=dict(bound="mfma", kernel="k" * 5000, achieved=1.0, peak=2.0, unit="TFLOP/s", frac=0.5, basis="b" * 5000,
                                   traffic=dict(bytes_per_launch=1e6, lanes_per_launch=20, profile="p" * 500), rocprof_in_chain=dict(avg_us=1.0, source="s" * 500)),
               cpu_baseline=dict(value=1.0, unit="u", cores=1, kind="port", sample="s" * 5000, cpu="c" * 500),
               also={("leg%d" % i): dict(value=float(i)) for i in range(40)})
    _check(bench.compact_line(fat), fat)


def test_emit_writes_the_full_file_and_one_stdout_line(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _round4_line()
    bench.emit(full)
    cap = capsys.readouterr()
    out_lines = [l for l in cap.out.splitlines() if l.strip()]
    assert len(out_lines) == 1 and len(out_lines[0]) <= bench.LINE_LIMIT
    line = json.loads(out_lines[0])
    assert line["full"] == os.path.join("gpurun_out", "bench_full.json")
    kept = json.load(open(tmp_path / "gpurun_out" / "bench_full.json"))
    assert kept["also"]["c3"]["roofline"] == full["also"]["c3"]["roofline"]   # nothing is lost: it moved
    assert "bench.py full result" in cap.err
