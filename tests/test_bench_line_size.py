"""bench.py's LAST stdout line is a record the driver parses (it keeps an 8 KB tail: round 5's 22 KB line came back `parsed: null`).
compact_line() is held here against the largest full object there is — round 5's committed 22 KB line, every optional leg and all its
prose present — and against a minimal one: the result round-trips through json, stays far below the limit, and carries the contract's
keys with a roofline and a cpu_baseline object."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _full():
    with open(os.path.join(ROOT, "profiles", "r05_bench_line.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_compact_line_of_the_largest_full_object_is_small_and_complete():
    full = _full()
    assert len(json.dumps(full)) > 20000   # (the stub really is the line that broke the parse)
    text = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT <= 6144, len(text)
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert set(line["config"]) == {"workload", "noise", "parallelism"}
    r = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_rocprof", "rocprof", "traffic", "traffic_source", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert set(r["rocprof"]) == {"source", "row", "avg_us"} and r["frac"] == full["roofline"]["frac"]
    for c in (line["cpu_baseline"], line["cpu_baseline_all_cores"], line["rbpf"]["cpu_baseline"]):
        assert set(c) == {"value", "unit", "cores", "kind", "sample"} and len(c["sample"]) <= 140
    rb = line["rbpf"]
    assert set(rb["modes"]) == {"reference_equal", "query_default"}
    for mo in rb["modes"].values():
        assert mo["particle_updates_per_s"] > 0 and mo["ms_per_scan"] > 0
    assert rb["roofline"]["kernel"].startswith("rbpf_raycast_box<") and rb["roofline"]["traffic_over_algorithmic"] > 1.0
    # no sentence-length strings anywhere in the record
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(x) for x in strings(line)) <= 160


def test_emit_prints_the_compact_record_last_and_keeps_everything_in_the_detail_file(tmp_path, monkeypatch):
    full = _full()
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    buf = io.StringIO()
    with redirect_stdout(buf):
        print("an earlier line")
        bench.emit(full)
    last = buf.getvalue().strip().splitlines()[-1]
    assert len(last) < bench.LINE_LIMIT and json.loads(last)["detail"] == "bench_detail.json"
    with open(tmp_path / "bench_detail.json") as f:
        assert json.load(f) == full


def test_a_minimal_full_object_still_gives_a_valid_record():
    full = {"metric": "MPPI rollouts/s", "value": 1.0, "unit": "rollouts/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.01,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "w", "noise": "n", "parallelism": "p"}, "roofline": None}
    line = json.loads(json.dumps(bench.compact_line(full)))
    assert line["roofline"] is None and line["cpu_baseline"] is None and "rbpf" not in line
