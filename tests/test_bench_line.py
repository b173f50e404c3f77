"""bench.py's final stdout line must stay parseable by the driver (the r05 line had grown to 42 KB,
the driver's bounded tail cut it, `parsed: null`).  Host logic only: no GPU, no oracle."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _full_record():
    # the complete r05 record (42 KB as one line): the very object whose one-line form the driver could not parse
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def test_compact_line_of_the_r05_record_fits_and_parses():
    B = _bench()
    out = _full_record()
    assert len(json.dumps(out)) > 30000
    line = B.compact_line(out, "bench_full.json")
    assert "\n" not in line and len(line.encode()) < 8192
    j = json.loads(line)
    for k in CONTRACT:
        assert k in j, k
    assert j["value"] == out["value"] and j["ms_per_step"] == out["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in j["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in j["cpu_baseline"], k
    assert j["config"]["workload"].startswith("configs[1]")
    assert "summary" in j and j["summary"]["k1_edges_per_s"] == out["value"]
    assert j["quality"]["rot_err_auc_at_5deg"] == out["quality"]["rot_err_auc_at_5deg"]


@pytest.mark.parametrize("blow", [10, 1000])
def test_compact_line_survives_a_record_that_keeps_growing(blow):
    B = _bench()
    out = _full_record()
    out["graphs"]["more"] = [{"all_repetitions_stage_s": list(range(200))} for _ in range(blow)]
    out["roofline"]["note"] = "x" * 5000
    out["cpu_baseline"]["sample"] = "y" * 5000
    out["config"]["exchange"] = "z" * 5000
    out["summary"] = dict(B.compact_summary(out), **{"extra_%d" % i: "w" * 200 for i in range(blow)})
    line = B.compact_line(out, "bench_full.json")
    assert len(line.encode()) < 8192
    j = json.loads(line)
    for k in CONTRACT:
        assert k in j, k
    assert j["roofline"]["frac"] == pytest.approx(out["roofline"]["frac"], rel=1e-5)


def test_emit_prints_the_line_last_and_writes_the_full_record(tmp_path, capsys, monkeypatch):
    B = _bench()
    out = _full_record()
    monkeypatch.setenv("PGI_BENCH_FULL", str(tmp_path / "bench_full.json"))
    B.emit(out)
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 1 and json.loads(lines[-1])["metric"] == out["metric"]
    assert json.load(open(tmp_path / "bench_full.json")) == out
    assert "bench_full.json" in cap.err
