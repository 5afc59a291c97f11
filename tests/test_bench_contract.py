"""The bench line's contract (one JSON object; metric/config of BASELINE.json; roofline + cpu_baseline objects), checked on
the committed line of the last GPU run (newest profiles/rNN/bench_final.json) and on bench.py's argument defaults."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    newest = sorted(r for r in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", r, "bench_final.json")))[-1]
    lines = [l for l in open(os.path.join(ROOT, "profiles", newest, "bench_final.json")).read().strip().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "mcts_simulations_per_sec" and d["unit"] == "simulations/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["trees_per_gpu"] * d["config"]["simulations"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    for name in ("roofline", "roofline_tree", "roofline_env") + (("roofline_env_hbm",) if newest >= "r02" else ()):
        r = d[name]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (name, k)
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["peak"] == 2500.0 and d["roofline_env"]["peak"] == 8000.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["value"] > 100 * c["value"]
    if newest >= "r02":   # round-2 additions: what the dtype string says, the reference-precision companion, the search-level bf16 check
        assert "tree" in d["dtype"] and "inference" in d["dtype"]
        assert d["fp32_inference"]["value"] > 0 and 0 < d["fp32_inference"]["roofline"]["frac"] < 1
        assert d["bf16_vs_fp32_search"]["top1_agreement"] >= 0.98
        assert c["ratio_to_reference"] > 0 and "valu" in d["roofline_env"] and d["roofline_env"]["valu"]["instructions_per_wave_step"] > 0


def test_bench_defaults_are_the_driver_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout, flag
