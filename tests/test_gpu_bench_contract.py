"""bench.py's output contract (one JSON line on stdout with the driver's keys + roofline + cpu_baseline) on a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600, check=True).stdout.decode().strip().splitlines()
    assert len(out) == 1, out
    return json.loads(out[0])


def test_extraction_bench_line():
    d = _run("--utts", "600", "--steps", "2", "--warmup", "1", "--cpu-budget", "2", "--parity-utts", "2")
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "utt/s" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 600 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "utt/s" and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)
    assert d["parity_rel_l2_max_vs_fp64_oracle"] < 1e-4
    assert d["roofline_pool"]["bound"] == "hbm" and 0 < d["roofline_pool"]["frac"] < 1.2


def test_fp32_and_training_bench_lines():
    d = _run("--utts", "300", "--steps", "1", "--warmup", "1", "--cpu-budget", "0", "--precision", "fp32")
    assert d["dtype"] == "f32" and d["roofline"]["peak"] == pytest.approx(157.3) and d["config"]["fused_pool"] is False
    t = _run("--mode", "train", "--steps", "3", "--warmup", "1")
    assert t["unit"] == "chunks/s" and t["value"] > 0 and t["last_loss"] > 0 and "AM-softmax" in t["metric"]
