"""Two ranks of the REAL extractor on the one GPU a test box has.  RCCL refuses two ranks on one device, so the group runs over
gloo (XVECTOR_DIST_BACKEND) and both ranks use cuda:0 (XVECTOR_DEVICE): everything of the N > 1 path except the transport --
the launcher, the side-thread group bring-up, scp line-range sharding, ark byte-range sharding, the stream mode, the one
gather, rank 0's writes -- against the bytes a single process writes.  The 8-GPU RCCL job itself is the driver's run."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, TWIN

pytestmark = pytest.mark.gpu


def _run(nproc, args, env):
    cmd = [sys.executable, "-m", "xvector_amd.launch", "--nproc", str(nproc), os.path.join(TWIN, "extract_embedding.py")] + args
    run = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    log = run.stdout.decode(errors="replace")
    assert run.returncode == 0, log[-3000:]
    return log


def test_two_rank_jobs_write_the_single_process_bytes(tmp_path, oracle_mod):
    import kaldi_io
    import models
    from xvector_amd import hiplib, synthetic, topology
    hiplib.require_gpu()
    topo = topology.get("ModelWithoutDropout")
    w = synthetic.trained_like(topo, 23, seed=5)
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=64, feat_dim=23), mdir, None)
    rng = np.random.default_rng(5)
    n = 3000
    lens = rng.integers(40, 400, size=n)
    lens[::97] = 11                                                   # rejected: shorter than min_chunk_size
    pool = [(rng.standard_normal((400, 23)) * 3.0).astype(np.float32) for _ in range(61)]
    ark, scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    with kaldi_io.TableWriter(ark, scp) as tw:
        for i in range(n):
            kaldi_io.write_mat(tw, pool[i % 61][:lens[i]], key="utt%05d" % i)
    base = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, os.environ.get("PYTHONPATH", "")]), XVECTOR_DEVICE="cuda:0",
                XVECTOR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "XV_FORCE_DIST"):
        base.pop(k, None)
    common = ["--use-gpu", "yes", "--min-chunk-size", "25", "--chunk-size", "10000", "--model-dir", mdir]

    def job(tag, nproc, rspec):
        out_ark, out_scp = str(tmp_path / (tag + ".ark")), str(tmp_path / (tag + ".scp"))
        log = _run(nproc, common + ["--feature-rspecifier", rspec, "--vector-wspecifier", "ark,scp:%s,%s" % (out_ark, out_scp)], base)
        return open(out_ark, "rb").read(), open(out_scp).read().replace(out_ark, "ARK"), log

    one_ark, one_scp, _ = job("one", 1, "scp:" + scp)
    kept = int((lens >= 25).sum())
    assert len(one_scp.splitlines()) == kept
    # (a) scp table: sharded by line range;  (b) the ark file itself: sharded by byte range;  (c) a pipe: every rank reads it
    for tag, rspec, marker in (("scp2", "scp:" + scp, None), ("ark2", "ark:" + ark, "bytes"), ("pipe2", "ark:cat %s |" % ark, None)):
        got_ark, got_scp, log = job(tag, 2, rspec)
        assert got_ark == one_ark and got_scp == one_scp, tag
        assert "Job wall clock:" in log
        if marker == "bytes":                          # both ranks report the byte range they took
            assert "rank 0 of 2: records 0.." in log and "rank 1 of 2: records " in log, log[-1500:]
    # (d) XVECTOR_SHARD_OUTPUT=files: one ark per rank + a concatenated scp, no process group at all: same keys, order, vectors
    files_env = dict(base, XVECTOR_SHARD_OUTPUT="files")
    f_ark, f_scp = str(tmp_path / "files.ark"), str(tmp_path / "files.scp")
    _run(2, common + ["--feature-rspecifier", "scp:" + scp, "--vector-wspecifier", "ark,scp:%s,%s" % (f_ark, f_scp)], files_env)
    want = list(kaldi_io.read_vec_flt_ark(io.BytesIO(one_ark)))
    got_f = list(kaldi_io.read_vec_flt_scp(f_scp))
    assert [k for k, _ in got_f] == [k for k, _ in want] and all(np.array_equal(a, b) for (_, a), (_, b) in zip(got_f, want))
    assert os.path.exists(f_ark + ".0") and os.path.exists(f_ark + ".1") and not os.path.exists(f_scp + ".0.part")
    # parity of a sample against the fp64 oracle (what all the jobs wrote)
    got = dict(kaldi_io.read_vec_flt_ark(io.BytesIO(one_ark)))
    for i in (0, 1, n // 2, n - 1):
        if lens[i] >= 25:
            ref = oracle_mod.embed_utterance(pool[i % 61][:lens[i]], w, topo, 25, 10000, np.float64)
            assert oracle_mod.rel_l2(got["utt%05d" % i], ref) < 1e-4


_TRAIN_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["XV_TEST_PKG"])
from xvector_amd import dist as xdist, synthetic, topology, trainer
rank, world = xdist.init_process_group()
assert world == 2
topo = topology.get("ModelWithoutDropout")
topo["layer_sizes"] = [64, 64, 64, 64, 96]; topo["embedding_sizes"] = [32, 32]
w = synthetic.trained_like(topo, 23, num_classes=10, seed=3)
tr = trainer.Trainer(w, topo, device="cuda:0", precision=os.environ["XV_TEST_PRECISION"])
for step in range(2):
    rng = np.random.default_rng(100 * step + rank)
    x = (rng.standard_normal((6, 80 + 10 * step, 23)) * 3).astype(np.float32)
    lab = rng.integers(0, 10, 6)
    tr.step(x, lab, 1e-3)
np.save(os.path.join(os.environ["XV_TEST_OUT"], "p%d.npy" % rank), tr.flat_p.cpu().numpy())
np.save(os.path.join(os.environ["XV_TEST_OUT"], "s%d.npy" % rank), tr.flat_moving.cpu().numpy())
xdist.finish_process_group()
'''


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_two_rank_training_steps_average_the_gradients(tmp_path, precision):
    """BASELINE configs[4] with N > 1: two ranks of the REAL trainer on cuda:0 (gloo transport), each on its own minibatches, two
    optimizer steps with the four bucketed all-reduces issued during the backward pass.  Both ranks end with bit-identical
    weights -- the ones a single process gets that runs both replicas itself, sums their gradients, halves them and applies
    Adam once -- while the batch-norm moving statistics stay per replica."""
    import math
    import torch
    from xvector_amd import hiplib, synthetic, topology, trainer
    hiplib.require_gpu()
    script = str(tmp_path / "train_worker.py")
    open(script, "w").write(_TRAIN_WORKER)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, os.environ.get("PYTHONPATH", "")]), XVECTOR_DEVICE="cuda:0",
               XVECTOR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", XV_TEST_PKG=PKG, XV_TEST_OUT=str(tmp_path),
               XV_TEST_PRECISION=precision)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "XV_FORCE_DIST"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, "-m", "xvector_amd.launch", "--nproc", "2", script], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=600)
    assert run.returncode == 0, run.stdout.decode(errors="replace")[-3000:]
    p0, p1 = np.load(str(tmp_path / "p0.npy")), np.load(str(tmp_path / "p1.npy"))
    s0, s1 = np.load(str(tmp_path / "s0.npy")), np.load(str(tmp_path / "s1.npy"))
    assert np.array_equal(p0, p1) and not np.array_equal(s0, s1)
    # the same two steps in ONE process: replicas A and B, gradients summed and halved exactly as Trainer.step does
    topo = topology.get("ModelWithoutDropout")
    topo["layer_sizes"] = [64, 64, 64, 64, 96]; topo["embedding_sizes"] = [32, 32]
    w = synthetic.trained_like(topo, 23, num_classes=10, seed=3)
    reps = [trainer.Trainer(w, topo, device="cuda:0", precision=precision) for _ in range(2)]
    for step in range(2):
        g = []
        for rank, tr in enumerate(reps):
            rng = np.random.default_rng(100 * step + rank)
            x = (rng.standard_normal((6, 80 + 10 * step, 23)) * 3).astype(np.float32)
            lab = rng.integers(0, 10, 6)
            tr.gradients(x, lab)
            g.append(tr.flat_g.clone())
        total = g[0] + g[1]
        for tr in reps:
            tr.flat_g.copy_(total)
            hiplib.axpy(tr.flat_g, tr.flat_g, 1.0 / 2 - 1.0)
            tr.t += 1
            lr_t = 1e-3 * math.sqrt(1.0 - trainer.ADAM_B2 ** tr.t) / (1.0 - trainer.ADAM_B1 ** tr.t)
            hiplib.adam(tr.flat_p, tr.flat_g, tr.flat_m, tr.flat_v, lr_t, trainer.ADAM_B1, trainer.ADAM_B2, trainer.ADAM_EPS)
            tr._packed = None
    torch.cuda.synchronize()
    assert np.array_equal(reps[0].flat_p.cpu().numpy(), p0)
    assert np.array_equal(reps[0].flat_moving.cpu().numpy(), s0) and np.array_equal(reps[1].flat_moving.cpu().numpy(), s1)
    alone = trainer.Trainer(w, topo, device="cuda:0", precision=precision)
    for step in range(2):
        rng = np.random.default_rng(100 * step)
        x = (rng.standard_normal((6, 80 + 10 * step, 23)) * 3).astype(np.float32)
        alone.step(x, rng.integers(0, 10, 6), 1e-3)
    assert not np.array_equal(alone.flat_p.cpu().numpy(), p0)       # (the exchange did something)
