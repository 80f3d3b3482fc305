"""The bench line committed under profiles/ carries every field the contract names."""
import json
import os

from conftest import ROOT


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r01_bench_200k_line.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "chunks/sec" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["config"]["chunks_per_step"] * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-6 * line["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
