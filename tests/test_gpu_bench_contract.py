"""bench.py's output contract (one JSON line on stdout with the driver's keys + roofline + cpu_baseline) on a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, **env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600, check=True, env=dict(os.environ, **env)).stdout.decode().strip().splitlines()
    assert len(out) == 1, out
    return json.loads(out[0])


def test_extraction_bench_line():
    d = _run("--utts", "600", "--steps", "2", "--warmup", "1", "--cpu-budget", "6", "--parity-utts", "2", "--e2e-utts", "700",
             XV_BENCH_REHEARSAL_UTTS="1500")                        # (the 8-rank job rehearsal at 8 x 1500 instead of 8 x 125 k utterances)
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
    assert isinstance(r["traffic_source"], str) and d["config"]["pair_kernel"] is True
    # the deployment-shaped CPU baseline (nj processes x 2 threads on the cores the container is granted) next to the faithful figure
    a = c["all_cores"]
    assert a["value"] > 0 and a["threads_per_process"] == 2 and a["cores"] == 2 * a["processes"] <= max(2, a["container_granted_cores"])
    assert c["reference_faithful_2_threads"] > 0 and c["value"] >= max(a["value"], c["reference_faithful_2_threads"]) * 0.999
    # exact-fp32 sub-record and the PCIe / parsing-inclusive ark -> ark rate ride on the same line; neither is `value`
    f = d["fp32_exact"]
    assert f["unit"] == "utt/s" and 0 < f["value"] < d["value"] and 0 < f["frac"] < 1 and f["peak_tflops"] == pytest.approx(157.3)
    assert r["algorithmic_rate_over_fp32_mfma_ceiling"] > 0 and not any(k.startswith("frac_of_fp32") for k in r)
    # the all-bf16x3 arithmetic (round 1's default, the twin of the f16bf8 default) in the same run, with its own parity figure
    assert d["config"]["precision"] == "f16bf8"
    b = d["bf16x3"]
    assert b["unit"] == "utt/s" and b["value"] > 0 and 0 < b["frac"] < 1 and b["parity_rel_l2_max_vs_fp64_oracle"] < 5e-5
    e = d["e2e_ark_to_ark"]
    assert e["utterances"] == 700 and e["vectors_written"] == 700 and e["value"] > 0
    assert e["fraction_of_resident_rate"] == pytest.approx(e["value"] / d["value"])
    assert e["run_time_accuracy_probe"]["probe_windows"] >= 1 and e["run_time_accuracy_probe"]["demoted"] is False
    # the arithmetic was admitted by the load-time probe, and the line says what it measured
    p = d["accuracy_probe"]
    assert p["probed"] and p["requested"] == p["selected"] == "f16bf8" and 0 < p["f16bf8_vs_bf16x3"] <= p["f16bf8_limit"] == 2e-5
    assert d["parity_utterances"] == 2 and d["rccl_ranks"] == 0 and "gather_ms" not in d
    # one rank's share of BASELINE configs[2] and configs[4], and the wall clock of one CLI worker job, on the same line
    v = d["config2_varlen"]
    assert v["unit"] == "utt/s" and v["value"] > 0 and v["frames_per_s"] > v["value"] * 25 and 0 < v["frac"] < 1 and v["batches"] > 10
    assert set(v["parity_rel_l2_vs_fp64_oracle"]) == {"T=25", "T=10000"} and max(v["parity_rel_l2_vs_fp64_oracle"].values()) < 1e-4
    t = d["train_step"]
    assert t["precision"] == "bf16x3" and 0 < t["ms_per_step"] < t["fp32"]["ms_per_step"] * 1.2 and t["chunks_per_s"] > 0
    assert 0 < t["mfma_time_over_time"] < 1 and 0 < t["fp32"]["mfma_time_over_time"] < 1 and t["last_loss"] < t["first_loss"]
    j = d["cli_job"]
    assert j["utterances"] == 700 and j["first_job_on_this_box"]["vectors_written"] == 700 and 0 < j["wall_s"] < 60
    assert j["breakdown_s"]["total"] <= j["wall_s"] and "import torch" in j["breakdown_s"] and "gather" in j["breakdown_s"]
    assert j["shard_files"]["vectors_written"] == 700 and "gather" not in j["shard_files"]["breakdown_s"]
    # the product default is ONE RCCL gather (the worker pre-loads RCCL's device code under `import torch`); RCCL without the pre-load
    # and the gloo gather are timed beside it
    assert j["transport"] == "nccl" and j["rccl_gather"]["transport"] == "nccl" and j["rccl_gather"]["vectors_written"] == 700
    assert j["rccl_gather"]["first_job"]["vectors_written"] == 700 and "RCCL pre-load (ok) joined after" in j["breakdown_s"]
    assert j["rccl_without_prewarm"]["transport"] == "nccl" and not any("pre-load" in k for k in j["rccl_without_prewarm"]["breakdown_s"])
    assert j["gloo_gather"]["transport"] == "gloo" and j["gloo_gather"]["vectors_written"] == 700
    # BASELINE configs[3] rehearsed at 8 ranks on this one GPU (scp line ranges, ONE gather, rank 0 writes every record), with the node's
    # per-rank job predicted piece by piece from it and from the single-rank job above
    h = j["rehearsal_8x125k"]
    assert h["ranks"] == 8 and h["utterances"] == 12000 == h["vectors_written"] and h["breakdown_s_rank0"]["gather"] >= 0
    assert set(h["predicted_breakdown_s"]["from_this_rehearsal"]) == {"tables opened", "gather", "write", "rename"}
    assert 0 < h["predicted_per_rank_job_s"] < 60 and h["predicted_8gpu_utt_per_s"] == pytest.approx(12000 / h["predicted_per_rank_job_s"])
    assert r["same_arithmetic"]["cli_job"]["rehearsal_8x125k"]["predicted_per_rank_job_s"] == pytest.approx(h["predicted_per_rank_job_s"], abs=1e-3)
    # what a record that keeps only the contract's objects still carries: the exact-fp32 legs (the reference's arithmetic), compactly
    sa = r["same_arithmetic"]
    assert sa["fp32_exact"]["utt_s"] == pytest.approx(f["value"], rel=1e-3) and 0 < sa["fp32_toomcook"]["frac_of_157.3TF_executed"] < 1
    assert sa["fp32_toomcook"]["algorithmic_over_peak"] > sa["fp32_toomcook"]["frac_of_157.3TF_executed"]
    assert sa["train_step_ms"]["bf16x3"] == pytest.approx(t["ms_per_step"], rel=1e-3) and sa["cli_job"]["transport"] == "nccl"
    assert 0 < d["fp32_toomcook"]["frac"] < 1 and d["fp32_toomcook"]["frac"] == d["fp32_toomcook"]["executed_frac"]
    assert "no exchange at N = 1" in d["config"]["parallelism"]


def test_two_ranks_through_the_self_launcher_on_one_gpu():
    """`python bench.py --gpus 2`, the driver's command form: bench.py starts its own two ranks (xvector_amd/launch.py), they form
    a group, run the sharded step with the gather and the max-over-ranks timing, and rank 0 alone prints ONE line.  A test box has
    one GPU and RCCL refuses two ranks on one device, so the ranks share it over gloo (XV_BENCH_SHARE_GPU / XVECTOR_DIST_BACKEND):
    the control flow of the N > 1 path on real kernels; the 8-GPU RCCL run itself is the driver's."""
    env = dict(os.environ, XV_BENCH_SHARE_GPU="1", XVECTOR_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--utts", "600", "--steps", "2", "--warmup", "1",
                          "--cpu-budget", "4", "--parity-utts", "4", "--e2e-utts", "0"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    out = run.stdout.decode().strip().splitlines()
    assert len(out) == 1, out
    d = json.loads(out[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["gather_ms"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 600 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]           # whole-job aggregate over both ranks
    assert d["parity_rel_l2_max_vs_fp64_oracle"] < 1e-4 and "TEST MODE" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d and "e2e_ark_to_ark" not in d and "config2_varlen" not in d        # N = 1 only
    assert d["with_ark_write"]["ark_mb"] > 2 * 600 * 2000 / 1e6                                          # both ranks' vectors reached rank 0


def test_two_training_ranks_through_the_self_launcher_on_one_gpu():
    """`python bench.py --mode train --gpus 2` (BASELINE configs[4], N > 1): two trainer ranks on the shared GPU over gloo -- bucketed
    gradient all-reduces during the backward pass, max-over-ranks timing, ONE line from rank 0 with the whole job's chunks per second."""
    env = dict(os.environ, XV_BENCH_SHARE_GPU="1", XVECTOR_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    out = run.stdout.decode().strip().splitlines()
    assert len(out) == 1, out
    d = json.loads(out[0])
    assert d["n_gpus"] == 2 and d["unit"] == "chunks/s" and d["scaling"] == "weak" and "x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 64 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] and d["last_loss"] > 0


def test_bench_at_the_per_rank_size_of_the_million_utterance_job():
    """BASELINE configs[3] gives every one of 8 ranks 125 k utterances: the resident-input step at that size (37.5 M frames,
    ~145 batches) on one GPU."""
    d = _run("--utts", "125000", "--steps", "1", "--warmup", "0", "--cpu-budget", "0", "--e2e-utts", "0", "--no-fp32-leg", "--no-extra-legs")
    assert d["config"]["utts_per_gpu"] == 125000 and d["config"]["batches_per_step"] > 100
    assert 36e6 < d["config"]["frames_per_gpu"] < 39e6 and d["value"] > 20000 and "fp32_exact" not in d and "e2e_ark_to_ark" not in d
    assert d["with_ark_write"]["ark_mb"] > 250


def test_fp32_and_training_bench_lines():
    d = _run("--utts", "300", "--steps", "1", "--warmup", "1", "--cpu-budget", "0", "--precision", "fp32")
    assert d["accuracy_probe"]["probed"] is False and "config2_varlen" not in d
    assert d["dtype"] == "f32" and d["roofline"]["peak"] == pytest.approx(157.3) and d["config"]["fused_pool"] is True
    t = _run("--mode", "train", "--steps", "3", "--warmup", "1")
    assert t["unit"] == "chunks/s" and t["value"] > 0 and t["last_loss"] > 0 and "AM-softmax" in t["metric"]
