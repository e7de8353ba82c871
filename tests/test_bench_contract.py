"""CPU: the committed bench line (profiles/r1_bench_latest.json, written by bench.py on an MI355X)
carries every field of the driver's contract, with consistent arithmetic."""
import json
import os

from conftest import ROOT


def test_recorded_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", "r1_bench_latest.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / measured launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    # PMC traffic is per launch and close to the algorithmic bytes for this streaming kernel
    assert r["traffic"] is None or 0.99 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1
    # value = elements of all ranks / wall time, consistent with ms_per_step
    per_step = d["config"]["elements_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
