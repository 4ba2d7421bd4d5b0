"""EIGHT ranks of the real extractor / bench / trainer on the ONE GPU a test box has ("Next round" 5a): a rehearsal of
the 8-GPU node's jobs (BASELINE configs[3], [4]; the reference's mechanism is nj jobs + `cat xvector.*.scp`, extract_xvectors.sh:63-95)
in the share-GPU mode of tests/test_gpu_two_ranks.py -- gloo transport (RCCL refuses two ranks on one device), everything else as
on the node: the launcher at 8 processes, the side-thread group bring-up, scp line-range and ark byte-range sharding into 8 parts,
ONE gather of 8 blocks, rank 0's writes, the bench's max-over-ranks timing.  So that the driver's first real 8-GPU run is not the
first time eight ranks of this code meet."""
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, TWIN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "XV_FORCE_DIST"):
        env.pop(k, None)
    return env


def test_bench_at_eight_ranks_prints_one_line():
    """`python bench.py --gpus 8`, the driver's command form for SCALE_rNN.json."""
    env = _clean_env(XV_BENCH_SHARE_GPU="1", XVECTOR_DIST_BACKEND="gloo")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--utts", "400", "--steps", "2", "--warmup", "1",
                          "--cpu-budget", "4", "--parity-utts", "4", "--e2e-utts", "0"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=900)
    assert run.returncode == 0, run.stderr.decode()[-3000:]
    out = run.stdout.decode().strip().splitlines()
    assert len(out) == 1, out
    d = json.loads(out[0])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["gather_ms"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 400 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]           # whole-job aggregate over all ranks
    assert d["parity_rel_l2_max_vs_fp64_oracle"] < 1e-4
    assert d["with_ark_write"]["ark_mb"] > 8 * 400 * 2000 / 1e6                                # every rank's vectors reached rank 0
    assert "cpu_baseline" not in d


def test_training_bench_at_eight_ranks():
    env = _clean_env(XV_BENCH_SHARE_GPU="1", XVECTOR_DIST_BACKEND="gloo")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "8", "--steps", "2", "--warmup", "1"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert run.returncode == 0, run.stderr.decode()[-3000:]
    out = run.stdout.decode().strip().splitlines()
    assert len(out) == 1, out
    d = json.loads(out[0])
    assert d["n_gpus"] == 8 and d["unit"] == "chunks/s" and "x8" in d["config"]["parallelism"] and d["last_loss"] == d["last_loss"]


def test_eight_rank_cli_jobs_write_the_single_process_bytes(tmp_path, oracle_mod):
    import kaldi_io
    import models
    from xvector_amd import hiplib, synthetic, topology
    hiplib.require_gpu()
    topo = topology.get("ModelWithoutDropout")
    w = synthetic.trained_like(topo, 23, seed=6)
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=64, feat_dim=23), mdir, None)
    rng = np.random.default_rng(6)
    n = 2005                                                          # not a multiple of 8
    lens = rng.integers(40, 400, size=n)
    lens[::89] = 11                                                   # rejected: shorter than min_chunk_size
    pool = [(rng.standard_normal((400, 23)) * 3.0).astype(np.float32) for _ in range(53)]
    ark, scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    with kaldi_io.TableWriter(ark, scp) as tw:
        for i in range(n):
            kaldi_io.write_mat(tw, pool[i % 53][:lens[i]], key="utt%05d" % i)
    base = _clean_env(PYTHONPATH=os.pathsep.join([PKG, os.environ.get("PYTHONPATH", "")]), XVECTOR_DEVICE="cuda:0",
                      XVECTOR_DIST_BACKEND="gloo")
    common = ["--use-gpu", "yes", "--min-chunk-size", "25", "--chunk-size", "10000", "--model-dir", mdir]

    def job(tag, nproc, rspec, env=base):
        out_ark, out_scp = str(tmp_path / (tag + ".ark")), str(tmp_path / (tag + ".scp"))
        cmd = [sys.executable, "-m", "xvector_amd.launch", "--nproc", str(nproc), os.path.join(TWIN, "extract_embedding.py")] + common + \
              ["--feature-rspecifier", rspec, "--vector-wspecifier", "ark,scp:%s,%s" % (out_ark, out_scp)]
        run = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        log = run.stdout.decode(errors="replace")
        assert run.returncode == 0, log[-3000:]
        return out_ark, out_scp, log

    one_ark, one_scp, _ = job("one", 1, "scp:" + scp)
    one_bytes, one_lines = open(one_ark, "rb").read(), open(one_scp).read().replace(one_ark, "ARK")
    assert len(one_lines.splitlines()) == int((lens >= 25).sum())
    for tag, rspec in (("scp8", "scp:" + scp), ("ark8", "ark:" + ark)):            # line-range / byte-range sharding into 8 parts
        a, s, log = job(tag, 8, rspec)
        assert open(a, "rb").read() == one_bytes and open(s).read().replace(a, "ARK") == one_lines, tag
        assert "Job wall clock:" in log
        if tag == "ark8":
            assert all(("rank %d of 8: records " % r) in log for r in range(8)), log[-2000:]
    # the reference's own output protocol at 8 ranks: one ark per rank + a concatenated scp, no process group
    fa, fs, _ = job("files8", 8, "scp:" + scp, dict(base, XVECTOR_SHARD_OUTPUT="files"))
    want = list(kaldi_io.read_vec_flt_ark(io.BytesIO(one_bytes)))
    got = list(kaldi_io.read_vec_flt_scp(fs))
    assert [k for k, _ in got] == [k for k, _ in want] and all(np.array_equal(x, y) for (_, x), (_, y) in zip(got, want))
    assert all(os.path.exists("%s.%d" % (fa, r)) for r in range(8)) and not [f for f in os.listdir(str(tmp_path)) if f.endswith(".part")]
    vec = dict(want)
    for i in (0, n // 3, n - 1):
        if lens[i] >= 25:
            assert oracle_mod.rel_l2(vec["utt%05d" % i], oracle_mod.embed_utterance(pool[i % 53][:lens[i]], w, topo, 25, 10000, np.float64)) < 1e-4
