"""The stdout line of bench.py, built on the CPU from a canned full result (tests/golden/bench_result_round5_full.json: the 20.8 KB
line round 5 printed, which the driver could not parse).  The line must stay well under the driver's 8 KB stdout tail, keep every key
the task contract names plus `roofline` and `cpu_baseline`, and lose nothing silently: the full result goes to bench_detail.json."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture()
def full():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_result_round5_full.json")))


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def test_line_of_the_round5_result_is_short_and_complete(bench, full, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert len(json.dumps(full)) > 20000                      # the canned result is the one that broke the driver's parse
    text = bench.emit(full)
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 8000 and len(text) < 6500, len(text)
    d = json.loads(text)
    for k in CONTRACT:
        assert k in d, k
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "model" not in d["config"] and "workload" in d["config"]
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-8 and abs(d["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-8
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "step_ms", "kernel", "kernel_avg_ms", "stages_ms_per_step", "traffic_source",
              "measured", "frac_of_measured", "frac_of_measured_mix", "dominant_kernel_share"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and r["traffic"] == full["roofline"]["traffic"]
    assert "traffic_by_kernel" not in r and "sq_per_step" not in text
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert d["parity_spot_err"] < 1e-5
    for k in ("secondary", "secondary_cfg2"):
        s = d[k]
        assert set(("value", "ms_per_step", "steps", "parity_spot_err")) <= set(s) and "frac" in s["roofline"] and "activation" not in s["config"] and "workload" not in s["config"]
    assert [e["name"] for e in d["secondary_end_to_end"]] == ["cfg3", "cfg2"] and all(e["pageable_value"] < e["value"] for e in d["secondary_end_to_end"])
    # nothing is lost: the detail file holds the full result
    detail = json.load(open(tmp_path / bench.DETAIL_FILE))
    assert detail == full and d["detail"] == bench.DETAIL_FILE


def test_an_oversized_line_drops_secondaries_instead_of_breaking_the_parse(bench, full, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full["cpu_baseline"]["sample"] = "x" * 5000               # someone grows a headline string: the secondaries give way, the headline does not
    text = bench.emit(full)
    d = json.loads(text)
    assert len(text) < bench.LINE_LIMIT and "dropped_for_length" in d and "roofline" in d and "cpu_baseline" in d
    assert json.load(open(tmp_path / bench.DETAIL_FILE))["secondary_cfg2"]["value"] > 0


def test_traffic_key_survives_a_refused_profile(bench, full, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full["roofline"]["traffic"] = None
    full["roofline"]["traffic_note"] = "newest profile of this shape predates the current device sources: re-profile"
    d = json.loads(bench.emit(full))
    assert d["roofline"]["traffic"] is None and "re-profile" in d["roofline"]["traffic_note"]
