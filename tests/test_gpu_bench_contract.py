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
    assert len(lines[0]) < 8000, len(lines[0])                # ... that fits the driver's 8 KB stdout tail whole
    d = json.loads(lines[0])
    full = json.load(open(os.path.join(ROOT, d["detail"])))   # everything else the run measured
    assert full["value"] == pytest.approx(d["value"], rel=1e-8) and full["config"]["name"] == "cfg1"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 300 and d["warmup"] == 100 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 480000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "measured", "frac_of_measured", "frac_of_measured_mix", "step_ms", "kernel", "kernel_avg_ms"):
        assert k in r, k
    assert r["traffic"] is not None or "traffic_note" in r
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    m = r["measured"]
    assert 3000 < m["write"] < m["read"] < 8000 and 3000 < m["copy"] < 8000, m        # this box's ceilings, GB/s: below the vendor peak, writes slower than reads
    assert abs(r["frac_of_measured"] - r["achieved"] / m["copy"]) < 1e-4 * r["frac_of_measured"]
    assert "kernel_frac" in r or "dominant_kernel_share" in r
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["parity_spot_err"] < 1e-5
    assert len(d["config"]["device_src_sha16"]) == 16 and len(d["config"]["build_flags_sha16"]) == 16
    act = full["config"]["activation"][0]
    assert set(("total_ms", "tables_ms", "upload_ms", "scratch_alloc_ms")) <= set(act)
    assert full["config"]["device_memory"]["total_bytes"] > full["config"]["device_memory"]["used_by_workload_bytes"] > 0
    e = d["secondary_end_to_end"][0]
    assert e["pinned"] is True and e["value"] > 0 and e["h2d_GBs"] > 0 and e["pageable_value"] > 0 and e["parity_spot_err"] < 1e-5
    assert full["secondary_end_to_end"][0]["pageable"]["pinned"] is False
    assert e["value"] < d["value"]                            # the PCIe-inclusive rate is a secondary: never the headline


@pytest.mark.gpu
def test_default_line_with_driver_style_flags_carries_the_secondaries():
    """`python bench.py --steps K --warmup W` as the driver runs it: the headline (cfg 3) with K and W as given, the 14-channel-input reading,
    cfg 2 (BASELINE configs[1]) with ITS OWN steady-state step counts — a millisecond step needs >= 25 ms of warm-up, K and W are the
    headline's — and the PCIe-inclusive legs; one JSON line, every parity spot check below 1e-5."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--cpu-sample-streams", "8"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    # the driver keeps an 8 KB tail of stdout: the whole line must fit in it (round 5's 20.8 KB line was never parsed), stderr stays short too
    assert len(lines[0]) < 8000 and len(p.stdout) < 8000 and len(p.stderr) < 4000, (len(lines[0]), len(p.stdout), len(p.stderr))
    d = json.loads(lines[0])
    assert "dropped_for_length" not in d and "cpu_baseline" in d and d["cpu_baseline"]["kind"] == "port"
    assert d["steps"] == 6 and d["warmup"] == 2 and "cfg3" in d["config"]["workload"] and d["parity_spot_err"] < 1e-5
    assert d["secondary"]["steps"] == 6 and d["secondary"]["parity_spot_err"] < 1e-5 and d["secondary"]["value"] < d["value"]
    c2 = d["secondary_cfg2"]
    assert c2["steps"] >= 100 and c2["warmup"] >= 40 and c2["config"]["name"] == "cfg2" and c2["parity_spot_err"] < 1e-5
    assert 0.15 < c2["roofline"]["frac"] < 0.40 and 0.12 < d["roofline"]["frac"] < 0.40
    c14 = d["secondary_cfg2_14ch"]                     # north_star's literal layout on the on-chip tile (round 6: the overlap-add form)
    assert c14["config"]["name"] == "cfg2-14ch" and c14["config"]["input_channels"] == 14 and c14["config"]["path"] == "fused overlap-add"
    assert c14["parity_spot_err"] < 1e-5 and 0.15 < c14["roofline"]["frac"] < 0.40 and c2["config"]["path"] == "fused overlap-add"
    assert [e["name"] for e in d["secondary_end_to_end"]] == ["cfg3", "cfg2"]
    # traffic: the committed PMC profile of THESE device sources, or an explicit refusal that says why — never a stale number
    assert d["roofline"]["traffic"] is not None or "profile" in d["roofline"]["traffic_note"], d["roofline"]
    full = json.load(open(os.path.join(ROOT, d["detail"])))
    assert full["secondary_cfg2"]["config"]["activation"] and full["roofline"]["frac_scope"]


@pytest.mark.gpu
def test_strong_scaling_and_equalizer_modes_run_on_the_gpu():
    """`--scaling strong` on one GPU (the job total on rank 0, every rate bucket present) and cfg 4's two equalizer forms, small: the line
    says which scaling / which form, the legs are the policy's kernels, parity spot checks hold."""
    def run(*argv):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) < 8000
        return json.loads(lines[0])
    d = run("--scaling", "strong", "--workload", "cfg5", "--streams", "24", "--seconds", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert d["scaling"] == "strong" and d["config"]["streams_total"] == 24 and d["config"]["streams_this_rank"] == 24
    assert [l["rate"] for l in d["config"]["legs"]] == [44100.0, 48000.0, 96000.0] and [l["streams"] for l in d["config"]["legs"]] == [8, 8, 8]
    d = run("--workload", "cfg4", "--streams", "16", "--seconds", "2", "--steps", "2", "--warmup", "1", "--cpu-sample-streams", "2")
    eq = d["config"]["equalizer"]
    assert "folded" in eq["mode"] and 11000 < eq["response_taps"] < 13000 and eq["tail_bound"] <= 1e-7 and d["parity_spot_err"] < 1e-5
    assert d["config"]["hrir_taps"] == 8640 + eq["response_taps"] - 1 and d["cpu_baseline"]["kind"] == "port"
    d = run("--workload", "cfg4", "--eq", "cascade", "--streams", "16", "--seconds", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary")
    assert "cascade" in d["config"]["equalizer"]["mode"] and d["config"]["hrir_taps"] == 8640 and "f64" in d["dtype"]
