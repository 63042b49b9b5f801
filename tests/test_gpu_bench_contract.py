"""The bench line's contract (the driver parses ONE JSON line of `python bench.py`): every key the task statement names, the two added
objects (`roofline`, `cpu_baseline`), and this repository's additions (`roofline.measured` — the ceilings measured on the box of the run,
SURVEY.md 8d —, `secondary_end_to_end` — the PCIe-inclusive rate through the host entry —, `config.activation`).  Run on the smallest
BASELINE configuration (cfg 1: one stereo stream through NeutralSH1.0) so that it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_carries_every_contract_field():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "300", "--warmup", "100", "--cpu-sample-streams", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]                  # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 300 and d["warmup"] == 100 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 480000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "measured", "frac_of_measured", "frac_of_measured_mix", "traffic_note"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    m = r["measured"]
    assert 3000 < m["write"] < m["read"] < 8000 and 3000 < m["copy"] < 8000, m        # this box's ceilings, GB/s: below the vendor peak, writes slower than reads
    assert abs(r["frac_of_measured"] - r["achieved"] / m["copy"]) < 1e-9
    assert "kernel_frac" in r or "dominant_kernel_share" in r
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["parity_spot_err"] < 1e-5
    act = d["config"]["activation"][0]
    assert set(("total_ms", "tables_ms", "upload_ms", "scratch_alloc_ms")) <= set(act)
    assert d["config"]["device_memory"]["total_bytes"] > d["config"]["device_memory"]["used_by_workload_bytes"] > 0
    e = d["secondary_end_to_end"][0]
    assert e["pinned"] is True and e["value"] > 0 and e["h2d_GBs"] > 0 and e["pageable"]["pinned"] is False and e["parity_spot_err"] < 1e-5
    assert e["value"] < d["value"]                            # the PCIe-inclusive rate is a secondary: never the headline
