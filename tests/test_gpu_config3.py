"""BASELINE configs[3] at PER-RANK scale on the one GPU a test box has: 1 M utterances over 8 ranks = 125 k utterances per
rank, pushed through the scp-sharded CLI path (`extract_embedding.py` under the package's launcher, the twin of
local/tf/extract_xvectors.sh:63-95) with a forced 1-rank RCCL group, `ark,scp` output.  Checks order, keys, byte framing,
scp offsets and parity of a sample against the fp64 oracle; the 8-rank run itself is the driver's (SCALE_rNN.json).
The JOB-LEVEL wall clock is part of the contract: the job runs twice -- the first RCCL communicator on a
fresh box loads librccl's device code cold (~3.5 s under the HIP runtime's lock) -- and the second run must finish within
JOB_WALL_LIMIT seconds; both print their breakdown (xvector_amd/jobclock.py).
Time-boxed: the subprocess is killed after 15 minutes; XV_TEST_SHARD_UTTS shrinks the shard (default 125000)."""
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, TWIN

pytestmark = pytest.mark.gpu

# round 2: 5.9 s warm.  Now ~3.1-3.4 s: 0.8 s `import torch` + ~1.0 s RCCL bring-up (its kernel load holds the HIP runtime's
# lock, so it only partly overlaps the model load) + 0.65 s extraction at 190-220 k utt/s + 0.25 s D2H and ark,scp write
JOB_WALL_LIMIT = 4.5


def test_one_rank_share_of_the_million_utterance_job(oracle_mod, tmp_path):
    import kaldi_io
    import models
    from xvector_amd import hiplib, synthetic, topology
    hiplib.require_gpu()
    n = int(os.environ.get("XV_TEST_SHARD_UTTS", "125000"))
    free = shutil.disk_usage(str(tmp_path)).free
    need = n * (300 * 23 * 4 + 2100) * 1.15
    if free < need:
        n = max(2000, int(n * free / need * 0.8))
    topo = topology.get("ModelWithoutDropout")
    w = synthetic.trained_like(topo, 23, seed=3)
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=64, feat_dim=23), mdir, None)
    # utterance i = the first T_i rows of one of 509 seeded matrices (distinct keys, T ~ U{200..400} like configs[1]); every
    # 5000th utterance is too short for min_chunk_size and must vanish from the output without disturbing the order
    rng = np.random.default_rng(20260928)
    pool = [(rng.standard_normal((400, 23)) * 3.0).astype(np.float32) for _ in range(509)]
    lens = synthetic.utterance_lengths(n, 200, 400, 77)
    lens[4999::5000] = 17
    keys = ["spk%05d-utt%07d" % (i % 9973, i) for i in range(n)]
    feats_ark, feats_scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    t0 = time.time()
    with kaldi_io.TableWriter(feats_ark, feats_scp) as tw:
        for i in range(n):
            kaldi_io.write_mat(tw, pool[i % 509][:lens[i]], key=keys[i])
    t_write = time.time() - t0
    ark, scp = str(tmp_path / "xvector.ark"), str(tmp_path / "xvector.scp")
    # one rank per GPU through the package's own launcher (xvector_amd/launch.py: the same RANK / WORLD_SIZE / MASTER_* contract as
    # torch.distributed.run without its elastic agent -- seconds of start-up that a job-level number has no use for)
    cmd = [sys.executable, "-m", "xvector_amd.launch", "--nproc", "1", os.path.join(TWIN, "extract_embedding.py"),
           "--use-gpu", "yes", "--min-chunk-size", "25", "--chunk-size", "10000", "--feature-rspecifier", "scp:" + feats_scp,
           "--vector-wspecifier", "ark,scp:%s,%s" % (ark, scp), "--model-dir", mdir]
    env = dict(os.environ, XV_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.environ.get("PYTHONPATH", "")]))
    walls = []
    for attempt in ("first job on this box (librccl / code objects cold)", "second job"):
        for f in (ark, scp):
            if os.path.exists(f):
                os.remove(f)
        t0 = time.time()
        run = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        t_cli = time.time() - t0
        walls.append(t_cli)
        log = run.stdout.decode(errors="replace")
        assert run.returncode == 0, log[-3000:]
        clock = [ln for ln in log.splitlines() if "Job wall clock:" in ln]
        assert clock, log[-2000:]
        print("\n%s: %.2f s wall = %.0f utt/s\n  %s" % (attempt, t_cli, n / t_cli, clock[-1].split("] ", 1)[-1]))
    if n == 125000:
        assert walls[1] <= JOB_WALL_LIMIT, "the 125 k-utterance shard took %.2f s of wall clock (limit %.1f)" % (walls[1], JOB_WALL_LIMIT)
    # the same shard with XVECTOR_SHARD_OUTPUT=files (the reference's own protocol: one ark per job + a concatenated scp,
    # extract_xvectors.sh:83-95): no process group, so nothing of RCCL's bring-up is in the job; same vectors, same order
    f_ark, f_scp = str(tmp_path / "xvector_f.ark"), str(tmp_path / "xvector_f.scp")
    cmd_f = [c.replace("ark,scp:%s,%s" % (ark, scp), "ark,scp:%s,%s" % (f_ark, f_scp)) for c in cmd]
    t0 = time.time()
    run = subprocess.run(cmd_f, env=dict(env, XVECTOR_SHARD_OUTPUT="files"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    t_files = time.time() - t0
    log_f = run.stdout.decode(errors="replace")
    assert run.returncode == 0, log_f[-3000:]
    clock = [ln for ln in log_f.splitlines() if "Job wall clock:" in ln]
    print("\nXVECTOR_SHARD_OUTPUT=files: %.2f s wall = %.0f utt/s\n  %s" % (t_files, n / t_files, clock[-1].split("] ", 1)[-1]))
    assert open(f_scp).read().replace(f_ark + ".0", "ARK") == open(scp).read().replace(ark, "ARK")
    assert open(f_ark + ".0", "rb").read() == open(ark, "rb").read()
    assert "Done %d and failed %d" % (n - n // 5000, n // 5000) in log, log[-2000:]
    # order, keys, framing: one FV record per surviving utterance, in input order, nothing else in the file
    kept = [i for i in range(n) if lens[i] >= 25]
    lines = open(scp).read().splitlines()
    assert len(lines) == len(kept)
    assert [ln.split(None, 1)[0] for ln in lines] == [keys[i] for i in kept]
    rec = lambda k: len(k) + 1 + 2 + 3 + 1 + 4 + 512 * 4          # key, space, \0B, "FV ", \4, int32 dim, payload (kaldi_io.py:309-343)
    assert os.path.getsize(ark) == sum(rec(keys[i]) for i in kept)
    assert not os.path.exists(ark + ".tmp.ark") and not os.path.exists(scp + ".tmp.scp")
    off = 0
    for j in (0, 1, len(kept) // 2, len(kept) - 1):               # scp offsets point just past "key " of their record
        off = sum(rec(keys[i]) for i in kept[:j])
        assert lines[j].split(None, 1)[1] == "%s:%d" % (ark, off + len(keys[kept[j]]) + 1)
    with open(ark, "rb") as f:                                     # first record's framing, byte for byte
        head = f.read(len(keys[kept[0]]) + 11)
    assert head == keys[kept[0]].encode() + b" \0BFV \x04" + (512).to_bytes(4, "little")
    # parity of a sample (through the scp, i.e. through the offsets) against the fp64 oracle
    sample = [kept[j] for j in np.linspace(0, len(kept) - 1, 6).astype(int)]
    want_lines = [lines[kept.index(i)] for i in sample]
    sub = str(tmp_path / "sample.scp")
    open(sub, "wt").write("\n".join(want_lines) + "\n")
    got = dict(kaldi_io.read_vec_flt_scp(sub))
    worst = 0.0
    for i in sample:
        ref = oracle_mod.embed_utterance(pool[i % 509][:lens[i]], w, topo, 25, 10000, np.float64)
        worst = max(worst, oracle_mod.rel_l2(got[keys[i]], ref))
    assert worst < 1e-4, worst                                     # the north star's bar; measured ~5e-6
    print("\nconfigs[3] shard: %d utterances (%.2f GB of features written in %.1f s), CLI wall %.1f s incl. interpreter + model load "
          "= %.0f utt/s, worst rel-L2 vs fp64 oracle %.2e" % (n, os.path.getsize(feats_ark) / 1e9, t_write, t_cli, n / t_cli, worst))
