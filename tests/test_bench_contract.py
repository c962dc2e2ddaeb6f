"""The bench line the driver parses (the committed run of the current build, profiles/): every field of the contract."""
import glob
import json
import os


def _latest():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    files = sorted(glob.glob(os.path.join(root, "r01_v*_bench.json")), key=lambda p: int(os.path.basename(p).split("_v")[1].split("_")[0]))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_fields():
    j = _latest()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(j[k], t), (k, type(j[k]))
    assert j["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert j["metric"] == "speech-tokens/sec" and j["higher_is_better"] is True and j["data"] == "synthetic"
    assert j["scaling"] in ("weak", "strong") and "workload" in j["config"] and "model" in j["config"]
    assert abs(j["value"] - 64 * 1e3 / j["ms_per_step"]) / j["value"] < 0.02      # whole-job tokens / wall time
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > r["bytes_per_step"]["weights_backbone"]
    assert abs(r["achieved"] * 1e9 * r["event_ms_per_step"] * 1e-3 - r["bytes_per_step"]["total"]) / r["bytes_per_step"]["total"] < 1e-6
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert c["unit"] == j["unit"]
