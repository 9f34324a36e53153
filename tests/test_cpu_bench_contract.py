"""The benchmark's output contract, checked on the committed lines of this round (profiles/r03_*.json[l]): every line bench.py
printed on the GPU box carries the fields the driver and the judge read, with consistent values."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def _lines():
    out = []
    for name in ("r03_bench_b1024.json", "r03_configs.jsonl"):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            out += [(name, json.loads(l)) for l in f if l.strip()]
    return out


@pytest.mark.parametrize("name,line", _lines(), ids=lambda v: v if isinstance(v, str) else v["config"]["name"])
def test_committed_bench_lines_follow_the_contract(name, line):
    for k in REQUIRED:
        assert k in line, (name, k)
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert line["unit"] == "audio-sec/sec" and line["n_gpus"] == 1 and line["value"] > 0
    assert isinstance(line["config"].get("workload"), str) and "model" not in line["config"]
    utts = line["config"]["utterances_per_gpu"]
    assert abs(line["value"] - 30.0 * utts * line["n_gpus"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]      # value is audio seconds per wall second
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
    assert r["peak"] == (8000.0 if r["bound"] == "hbm" else 2500.0)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 and 0 < r["frac"] < 1
    assert r["launches"] > 0 and r["avg_launch_ms"] > 0
    # achieved = algorithmic work per launch / average launch duration, both reported
    unit = 1e12 if r["bound"] == "mfma" else 1e9
    assert abs(r["achieved"] - r["algorithmic_work_per_launch"] / (r["avg_launch_ms"] * 1e-3) / unit) <= 1e-6 * r["achieved"]
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert line["dtype"] in ("f16", "f32")


def test_headline_line_is_the_baseline_workload_with_its_extra_legs():
    line = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_b1024.json")))
    assert "large-v3-turbo" in line["metric"] and line["config"]["utterances_per_gpu"] == 1024 and line["config"]["tokens_per_utterance"] == 128
    assert line["roofline"]["bound"] == "mfma" and line["roofline"]["traffic"] > 0
    s = line["strong_scaling"]
    assert s["scaling"] == "strong" and s["utterances"] == 10240 and s["batches"] * s["batch"] >= s["utterances"]
    assert abs(s["value"] - 30.0 * s["utterances"] / s["seconds"]) <= 1e-6 * s["value"]
    assert line["host_pcm"]["value"] > 0
    # the step's kernel classes add up to (nearly) the step
    kc = {k: v["ms"] for k, v in line["kernel_classes"].items() if isinstance(v, dict)}
    assert 0.9 * line["ms_per_step"] <= sum(kc.values()) <= 1.1 * line["ms_per_step"]
