"""N>1 path on CPU: world_size-2 gloo processes run the sharding + single-gather logic of
xvector_amd.dist with a stand-in per-utterance function (the GPU forward itself is covered by -m gpu)."""
import os
import subprocess
import time
import sys
import textwrap

import numpy as np
import pytest

from conftest import PKG
from xvector_amd import dist as xdist


def test_gather_transport_follows_the_payload(monkeypatch):
    """dist.gather_backend: the transport of a job whose one exchange is the final gather -- forced by XVECTOR_DIST_BACKEND, gloo
    without a GPU, and with one gloo up to XVECTOR_HOST_GATHER_MAX_MB of vectors (default 512), RCCL above; a pending asynchronous
    bring-up takes the choice as long as its start mark has not fired."""
    import torch
    monkeypatch.delenv("XVECTOR_DIST_BACKEND", raising=False)
    monkeypatch.delenv("XVECTOR_HOST_GATHER_MAX_MB", raising=False)
    assert xdist.gather_backend(10 ** 12) == "gloo"                 # no GPU here
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    assert xdist.gather_backend(50000 * 513 * 4) == "gloo" and xdist.gather_backend(10 ** 6 * 513 * 4) == "nccl"
    assert xdist.gather_backend(512 * 10 ** 6) == "gloo" and xdist.gather_backend(512 * 10 ** 6 + 1) == "nccl"
    monkeypatch.setenv("XVECTOR_HOST_GATHER_MAX_MB", "0")
    assert xdist.gather_backend(4) == "nccl"
    monkeypatch.setenv("XVECTOR_DIST_BACKEND", "gloo")
    assert xdist.gather_backend(10 ** 12) == "gloo"
    monkeypatch.delenv("XVECTOR_DIST_BACKEND")
    monkeypatch.delenv("XVECTOR_HOST_GATHER_MAX_MB")
    import threading
    go = threading.Event()
    monkeypatch.setattr(xdist, "_ASYNC", dict(thread=object(), box={}, go=go))
    xdist.set_gather_payload(1000)
    assert xdist._ASYNC["backend"] == "gloo"
    xdist.set_gather_payload(10 ** 10)
    assert xdist._ASYNC["backend"] == "nccl"
    go.set()                                                        # the bring-up has started: too late to change its transport,
    with pytest.raises(RuntimeError, match="bring-up has started"):  # and said loudly (the ranks could otherwise disagree and hang)
        xdist.set_gather_payload(1000)
    assert xdist._ASYNC["backend"] == "nccl"
    monkeypatch.setenv("XVECTOR_RCCL_PREWARMED", "1")              # the CLI worker pre-loaded RCCL's device code: RCCL whatever the size
    from xvector_amd import rccl_prewarm
    assert rccl_prewarm.ENV_FLAG == "XVECTOR_RCCL_PREWARMED" and xdist.gather_backend(4) == "nccl"


def test_partition_lpt_is_balanced_and_deterministic():
    lens = np.random.default_rng(0).integers(25, 10001, size=1000)
    for world in (1, 2, 4, 8):
        shards = xdist.partition_lpt(lens, world)
        allidx = np.concatenate(shards)
        assert sorted(allidx) == list(range(1000))
        loads = [int(lens[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= lens.max()
        assert all(np.array_equal(a, b) for a, b in zip(shards, xdist.partition_lpt(lens, world)))
        assert all(np.all(np.diff(s) > 0) for s in shards if len(s) > 1)
    assert [len(s) for s in xdist.partition_lpt([5, 5, 5], 8)] == [1, 1, 1, 0, 0, 0, 0, 0]


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from xvector_amd import dist as xdist
    rank, world = xdist.init_process_group("gloo")
    lens = np.random.default_rng(3).integers(25, 400, size=int(os.environ.get("XV_TEST_UTTS", "37")))
    dim = 16
    def fake_extract(idx):      # stand-in for the GPU forward: row i = f(utterance i)
        return torch.stack([torch.full((dim,), float(i)) + torch.arange(dim) * float(lens[i]) for i in idx]) if len(idx) else torch.zeros((0, dim))
    out = xdist.sharded_extract(lens, fake_extract, dim, torch.device("cpu"))
    if rank == 0:
        want = torch.stack([torch.full((dim,), float(i)) + torch.arange(dim) * float(lens[i]) for i in range(len(lens))])
        assert torch.equal(out, want), "gathered result is not in input order"
        print("GATHER_OK", world)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("world,utts", [(2, 37), (3, 37), (8, 37), (8, 5)])
def test_two_rank_gloo_sharded_gather(tmp_path, world, utts):
    """(also at the driver's largest launch, 8 ranks -- and with fewer utterances than ranks: empty shards take part in the ONE
    padded gather like the others)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % PKG)
    port = 29000 + (os.getpid() % 2000) + 7 * world + utts
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   XV_TEST_UTTS=str(utts), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK %d" % world in outs[0]


SKIP_WORKER = textwrap.dedent("""
    import os, sys, logging, types
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from xvector_amd import dist as xdist
    import models
    rank, world = xdist.init_process_group("gloo")

    class Stepper(object):            # stands in for trainer.Trainer: one collective per step, like the gradient all-reduce
        device = "cpu"
        def __init__(self): self.steps = []
        def step(self, x, labels, lr, dp, seed):
            t = torch.tensor([float(x[0, 0, 0])]); dist.all_reduce(t)
            self.steps.append(t.item())
            return 1.0, 0.5
        def export(self): raise AssertionError("save_model is off")

    class Loader(object):             # rank 1 has no batch at index 1 (None) and times out at index 3
        count = 5
        def __init__(self): self.i = -1
        def pop(self, timeout=30):
            import queue
            self.i += 1
            if rank == 1 and self.i == 1: return None, None
            if rank == 1 and self.i == 3: raise queue.Empty()
            return np.full((2, 7, 3), float(self.i + 1), np.float32), np.zeros(2, np.int32)

    tr = Stepper()
    models.Model._trainer = lambda self, d, lg, first_batch=None: tr
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%%(levelname)s %%(message)s")
    args = types.SimpleNamespace(learning_rate=1e-3, print_interval=2, dropout_proportion=0.0, random_seed=0, input_dir="unused",
                                 output_dir="unused", save_model=False)
    models.Model().train_one_iteration(Loader(), args, logging.getLogger("skip"))
    # both ranks stepped on indices 0, 2, 4 only, and every step paired batches of the SAME index (sum = 2 * (index + 1))
    assert tr.steps == [2.0, 6.0, 10.0], tr.steps
    dist.barrier()
    print("SKIP_OK", rank)
    dist.destroy_process_group()
""")


def test_two_rank_gloo_training_skips_minibatches_together(tmp_path):
    """A rank whose loader times out or yields None must not skip a step on its own: the gradient all-reduce of the other
    rank would pair with a later step (or hang).  Model.train_one_iteration agrees on the skip with one MIN all-reduce."""
    from conftest import TWIN
    script = tmp_path / "skip_worker.py"
    script.write_text(SKIP_WORKER % (PKG, TWIN))
    port = 33000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "SKIP_OK 0" in outs[0] and "SKIP_OK 1" in outs[1]
    assert "skipped: another rank of the group has no batch" in outs[0]
    assert "batch_data is None for the minibatch index 1" in outs[1] and "Timeout reach when reading the minibatch index 3" in outs[1]
    assert "Overall average objective function is -0.6000 over 6 segments." in outs[0]      # 3 steps of loss 1.0 over 5 planned


DRIVER_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import train_dnn

    class FakeModel(object):            # stands in for the GPU trainer: collectives in every step, like Trainer.step
        def build_model(self, num_classes, feat_dim, out, logger=None):
            os.makedirs(out); open(os.path.join(out, "model.meta"), "wt").write("init"); open(os.path.join(out, "done"), "wt").write("done")
        def train_one_iteration(self, loader, args, logger):
            steps = 0
            for _ in range(loader.count):
                data, lab = loader.pop()
                assert data is not None
                t = torch.tensor([1.0]); dist.all_reduce(t); assert t.item() == dist.get_world_size()
                steps += 1
            logger.info("Overall average objective function is %%.4f over %%d segments." %% (-1.0, steps))
            assert args.save_model == (dist.get_rank() == 0)
            if args.save_model:
                os.makedirs(args.output_dir)
                open(os.path.join(args.output_dir, "model.meta"), "wt").write("from %%s steps=%%d lr=%%.6f seed=%%d" %% (
                    os.path.basename(args.input_dir), steps, args.learning_rate, args.random_seed))
                open(os.path.join(args.output_dir, "done"), "wt").write("done")
        def eval(self, loader, input_dir, use_gpu, logger):
            logger.info("Overall average loss is 0.5000 over 8 segments. Also, the overall average accuracy is 0.7500.")

    train_dnn.models.Model = FakeModel
    train_dnn.models.ModelWithoutDropout = FakeModel
    train_dnn.main(sys.argv[1:])
    print("DRIVER_OK")
""")


def test_two_rank_gloo_training_driver(tmp_path):
    """train_dnn.py under a 2-rank gloo group with a stand-in model: archive assignment per rank (train_dnn.py:246-249), the
    step count capped to the shortest archive of the iteration, a collective per step on both ranks, rank 0 alone writes the
    models, per-job logs, diagnostics spread over the ranks, model_final + accuracy.report."""
    import examples_io
    from conftest import TWIN
    egs, exp = str(tmp_path / "egs"), str(tmp_path / "exp")
    os.makedirs(os.path.join(egs, "info")); os.makedirs(os.path.join(egs, "temp"))
    open(os.path.join(egs, "info", "feat_dim"), "wt").write("5\n")
    open(os.path.join(egs, "info", "num_archives"), "wt").write("4\n")
    rng = np.random.default_rng(0)
    counts = {1: 3, 2: 2, 3: 3, 4: 4}
    with open(os.path.join(egs, "temp", "archive_minibatch_count"), "wt") as fid:
        for a, c in counts.items():
            examples_io.write_egs_tar(os.path.join(egs, "egs.%d.tar" % a), [rng.standard_normal((4, 10 + a, 5)) for _ in range(c)],
                                      rng.integers(0, 3, (c, 4)))
            fid.write("%d %d\n" % (a, c))
    for name in ("valid_egs.1.tar", "train_subset_egs.1.tar"):
        examples_io.write_egs_tar(os.path.join(egs, name), [rng.standard_normal((4, 9, 5))], rng.integers(0, 3, (1, 4)))
    script = tmp_path / "driver_worker.py"
    script.write_text(DRIVER_WORKER % (PKG, TWIN))
    port = 31000 + (os.getpid() % 2000)
    flags = ["--tf-model-class", "ModelWithoutDropout", "--dir", exp, "--egs-dir", egs, "--num-targets", "3", "--minibatch-size", "4",
             "--num-epochs", "1", "--cleanup", "false", "--random-seed", "7"]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)] + flags, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("DRIVER_OK" in o for o in outs)
    # 4 archives / 2 ranks = 2 iterations; iteration 0: archives 1,2 -> min(3,2) steps; iteration 1: archives 3,4 -> min(3,4)
    m1 = open(os.path.join(exp, "model_1", "model.meta")).read()
    m2 = open(os.path.join(exp, "model_2", "model.meta")).read()
    assert m1.startswith("from model_0 steps=2 ") and m1.endswith("seed=7"), m1
    assert m2.startswith("from model_1 steps=3 ") and m2.endswith("seed=8"), m2
    lr1, lr2 = float(m1.split("lr=")[1].split()[0]), float(m2.split("lr=")[1].split()[0])
    assert abs(lr1 - 2 * 0.0003) < 1e-9 and abs(lr2 - 2 * 0.00003) < 1e-9              # ze_utils.py:111-120 with num_jobs = 2
    logs = sorted(os.listdir(os.path.join(exp, "log")))
    assert [l for l in logs if l.startswith("train.")] == ["train.0.1.log", "train.0.2.log", "train.1.1.log", "train.1.2.log"]
    assert "over 2 segments" in open(os.path.join(exp, "log", "train.0.2.log")).read()
    assert "compute_prob_valid.1.log" in logs and "compute_prob_train_subset.1.log" in logs
    assert os.readlink(os.path.join(exp, "model_final")) == "model_2"
    rep = open(os.path.join(exp, "accuracy.report")).read().splitlines()
    assert len(rep) == 4 and rep[1].split("\t")[0] == "0" and rep[1].split("\t")[2:] == ["0.5", "0.5", "0", "0.75", "0.75", "0"]


EMBED_WORKER = textwrap.dedent("""
    import io, logging, os, sys, types
    sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
    import numpy as np, torch.distributed as dist
    import kaldi_io, models
    from xvector_amd import engine

    class FakeExtractor(object):          # stand-in for the GPU extractor: vector = f(utterance), None below min_chunk_size
        def __init__(self, model, min_chunk_size, chunk_size, **kw):
            self.min = min_chunk_size
            self.stats = dict(batches=0, chunks=0, frames=0, rows=0)
        def extract(self, mats):
            self.stats["frames"] += sum(m.shape[0] for m in mats)
            return [None if m.shape[0] < self.min else np.concatenate([m.mean(0), [m.shape[0]]]).astype(np.float32) for m in mats]
        def submit(self, mats, addrs=None):
            return self.extract(mats)
        def finish(self, h, as_array=False):
            valid = np.array([v is not None for v in h], bool)
            full = np.zeros((len(h), 6), np.float32)
            for i, v in enumerate(h):
                if v is not None:
                    full[i] = v
            return (full, valid) if as_array else h

    def fake_load(self, sess, input_dir, logger):
        self.device_model = types.SimpleNamespace(device="cpu", embed_dim=6, feat_dim=5)
    engine.Extractor = FakeExtractor
    models.Model.load_model = fake_load
    models.Model.window_frames = 400          # several windows, still ONE gather
    models.Model.arena_bytes = models.Model.first_arena_bytes = 4096      # (in-place reader: one arena = one window)

    import torch.distributed as _d
    _calls = []
    def _count(name):
        real = getattr(_d, name)
        def wrapped(*a, **k):
            _calls.append(name)
            return real(*a, **k)
        setattr(_d, name, wrapped)
    for _n in ("gather", "all_gather", "all_gather_object", "gather_object", "broadcast", "broadcast_object_list", "all_reduce",
               "scatter", "all_to_all", "send", "recv", "reduce"):
        _count(_n)
    log = logging.getLogger("w"); log.addHandler(logging.NullHandler())
    out = io.BytesIO()
    with open(sys.argv[1], "rb") as fin:
        src = fin
        if len(sys.argv) > 3 and sys.argv[3] == "pipe":          # a stream nobody can seek in or split (stands in for a pipe)
            src = io.BufferedReader(io.BytesIO(fin.read()))
        models.Model().make_embedding(src, out, "unused", 10, -1, True, log)
    rank = int(os.environ.get("RANK", "0"))
    open(sys.argv[2] + ".%%d" %% rank, "wb").write(out.getvalue())
    print("EMBED_OK collectives=%%s" %% ",".join(_calls))
    print("RANGE_BYTES=%%d" %% kaldi_io.FileRange.bytes_read)
""")


@pytest.mark.parametrize("mode", ["file", "pipe", "file_with_a_double_record", "pipe_exchange_every_2_windows", "file_of_compressed_matrices"])
def test_two_rank_gloo_make_embedding_equals_single_process(tmp_path, mode):
    """Model.make_embedding under a 2-rank gloo group (stand-in extractor), ONE gather at the very end (however many windows),
    rank 0 alone writes -- exactly the bytes a single process writes (input order restored, rejected utterances dropped).
    * ``pipe``: a stream nobody can split -- every rank reads it and extracts its frame-balanced shard of each window;
    * ``file``: a seekable ark file is split by BYTE RANGES (one header-only index pass per rank, no collective for it): each
      rank reads only its own records' bytes -- together exactly the file, neither of them more than its share;
    * the same for a file of CompressedMatrix records (Kaldi's default feature format): indexed by their headers, decoded natively;
    * a file holding a record the index pass does not take (a double-precision matrix) falls back to the stream mode;
    * XVECTOR_EXCHANGE_WINDOWS=2: the stream mode exchanges (and rank 0 writes) every two windows instead of once at the end."""
    import kaldi_io
    from conftest import TWIN
    rng = np.random.default_rng(2)
    lens = [30, 5, 80, 12, 0, 200, 45, 9, 60, 33, 150, 10, 71]
    ark = tmp_path / "feats.ark"
    with open(ark, "wb") as f:
        for i, t in enumerate(lens):
            m = rng.standard_normal((t, 5)).astype(np.float32)
            if mode == "file_of_compressed_matrices" and t > 0:
                from fixture_inputs import encode_cm_record
                f.write(encode_cm_record("utt%02d" % i, m))
                continue
            kaldi_io.write_mat(f, m.astype(np.float64) if (mode == "file_with_a_double_record" and i == 6) else m, key="utt%02d" % i)
    script = tmp_path / "embed_worker.py"
    script.write_text(EMBED_WORKER % (PKG, TWIN, os.path.dirname(PKG)))
    extra = ["pipe"] if mode.startswith("pipe") else []
    single = subprocess.run([sys.executable, str(script), str(ark), str(tmp_path / "single")] + extra, stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}, timeout=240)
    assert single.returncode == 0, single.stdout.decode()
    port = 33000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if mode == "pipe_exchange_every_2_windows":
            env["XVECTOR_EXCHANGE_WINDOWS"] = "2"
        procs.append(subprocess.Popen([sys.executable, str(script), str(ark), str(tmp_path / "dist")] + extra, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    want = open(str(tmp_path / "single.0"), "rb").read()
    got = {k: v for k, v in kaldi_io.read_vec_flt_ark(str(tmp_path / "dist.0"))}
    assert open(str(tmp_path / "dist.0"), "rb").read() == want
    assert open(str(tmp_path / "dist.1"), "rb").read() == b""                      # the non-root rank writes nothing
    if mode == "pipe_exchange_every_2_windows":      # XVECTOR_EXCHANGE_WINDOWS: a bounded stash -- more gathers, still nothing but gathers
        calls = [o.split("EMBED_OK collectives=")[1].split()[0].split(",") for o in outs]
        assert calls[0] == calls[1] and len(calls[0]) >= 2 and set(calls[0]) == {"gather"}, calls
    else:
        assert all("EMBED_OK collectives=gather\n" in o for o in outs), outs     # the whole job: exactly one data-path collective
    assert list(got) == ["utt%02d" % i for i, t in enumerate(lens) if t >= 10]
    assert got["utt05"][-1] == 200.0
    ranged = [int(o.split("RANGE_BYTES=")[1].split()[0]) for o in outs]
    if mode in ("file", "file_of_compressed_matrices"):
        size = os.path.getsize(str(ark))
        assert sum(ranged) == size and 0 < min(ranged) and max(ranged) <= 0.7 * size, (ranged, size)
    else:
        assert ranged == [0, 0]                                 # the whole stream went through every rank's reader


CLI_WORKER = textwrap.dedent("""
    import os, sys, types
    sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
    import numpy as np
    import models, extract_embedding
    from xvector_amd import engine

    class FakeExtractor(object):          # stand-in for the GPU extractor (see test_two_rank_gloo_make_embedding...)
        def __init__(self, model, min_chunk_size, chunk_size, **kw):
            self.min = min_chunk_size
            self.stats = dict(batches=0, chunks=0, frames=0, rows=0)
        def extract(self, mats):
            self.stats["frames"] += sum(m.shape[0] for m in mats)
            return [None if m.shape[0] < self.min else np.concatenate([m.mean(0), [m.shape[0]]]).astype(np.float32) for m in mats]
        def submit(self, mats, addrs=None):
            return self.extract(mats)
        def finish(self, h, as_array=False):
            valid = np.array([v is not None for v in h], bool)
            full = np.zeros((len(h), 6), np.float32)
            for i, v in enumerate(h):
                if v is not None:
                    full[i] = v
            return (full, valid) if as_array else h

    def fake_load(self, sess, input_dir, logger):
        self.device_model = types.SimpleNamespace(device="cpu", embed_dim=6, feat_dim=5)
    engine.Extractor = FakeExtractor
    models.Model.load_model = fake_load
    models.Model.window_frames = 300

    import torch.distributed as _d
    _calls = []
    def _count(name):
        real = getattr(_d, name)
        def wrapped(*a, **k):
            _calls.append(name)
            return real(*a, **k)
        setattr(_d, name, wrapped)
    for _n in ("gather", "all_gather", "all_gather_object", "gather_object", "broadcast", "broadcast_object_list", "all_reduce",
               "scatter", "all_to_all", "send", "recv", "reduce"):
        _count(_n)
    extract_embedding.main(sys.argv[1:])
    print("CLI_OK collectives=%%s" %% ",".join(_calls))
""")


def test_two_rank_gloo_cli_shards_an_scp_table(tmp_path):
    """extract_embedding.py under a 2-rank group with an 'scp:' feature table: every rank reads only its line range, one
    gather at the end, rank 0 writes ark + scp -- identical files to the single-process run (order, rejected keys, offsets)."""
    import kaldi_io
    from conftest import TWIN
    rng = np.random.default_rng(4)
    lens = [30, 5, 80, 12, 0, 200, 45, 9, 60, 33, 150, 10, 71, 8, 90]
    ark, scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    with kaldi_io.TableWriter(ark, scp) as w:
        for i, t in enumerate(lens):
            kaldi_io.write_mat(w, rng.standard_normal((t, 5)).astype(np.float32), key="utt%02d" % i)
    mdir = tmp_path / "model"; mdir.mkdir(); (mdir / "model.meta").write_text("x"); (mdir / "done").write_text("done")
    script = tmp_path / "cli_worker.py"
    script.write_text(CLI_WORKER % (PKG, TWIN, os.path.dirname(PKG)))

    def flags(tag):
        return ["--min-chunk-size", "10", "--feature-rspecifier", "scp:" + scp, "--model-dir", str(mdir),
                "--vector-wspecifier", "ark,scp:%s,%s" % (tmp_path / (tag + ".ark"), tmp_path / (tag + ".scp"))]
    base_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, str(script)] + flags("one"), env=base_env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert one.returncode == 0, one.stdout.decode()
    port = 35000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(base_env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)] + flags("two"), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert open(str(tmp_path / "two.ark"), "rb").read() == open(str(tmp_path / "one.ark"), "rb").read()
    s1 = open(str(tmp_path / "one.scp")).read().replace("one.ark", "X")
    s2 = open(str(tmp_path / "two.scp")).read().replace("two.ark", "X")
    assert s1 == s2 and len(s1.splitlines()) == sum(t >= 10 for t in lens)
    assert not os.path.exists(str(tmp_path / "two.ark.tmp.ark"))
    # north star: "a single RCCL gather at the end" -- no object collectives for counts or keys (the line ranges are deterministic)
    assert all("CLI_OK collectives=gather\n" in o for o in outs), outs
    # XVECTOR_SHARD_OUTPUT=files: the reference's own protocol (extract_xvectors.sh:83-95) -- every rank writes its own ark while
    # it extracts, rank 0 concatenates the scp parts; no process group, no collective at all; same keys, order and vectors
    port += 1
    # ... and what an earlier job with the same output paths left behind when it died before its concatenation: a complete-looking part
    # of rank 1 (these ranks share no job token).  Rank 0 must wait for THIS job's part, not concatenate the old one.
    stale = str(tmp_path / "three.scp.1.part")
    with open(stale, "wt") as f:
        f.write("ghost %s:7\n" % (tmp_path / "nowhere.ark.1"))
    os.utime(stale, (1e9, 1e9))
    procs = []
    for r in (0, 1):
        env = dict(base_env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   XVECTOR_SHARD_OUTPUT="files")
        procs.append(subprocess.Popen([sys.executable, str(script)] + flags("three"), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        if r == 0:
            time.sleep(1.5)             # rank 0 is at its wait loop before rank 1 even starts
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("CLI_OK collectives=\n" in o for o in outs), outs
    want = list(kaldi_io.read_vec_flt_scp(str(tmp_path / "one.scp")))
    got = list(kaldi_io.read_vec_flt_scp(str(tmp_path / "three.scp")))
    assert [k for k, _ in got] == [k for k, _ in want] and all(np.array_equal(a, b) for (_, a), (_, b) in zip(got, want))
    lines = open(str(tmp_path / "three.scp")).read().splitlines()
    assert {ln.split()[1].rsplit(":", 1)[0] for ln in lines} == {str(tmp_path / "three.ark.0"), str(tmp_path / "three.ark.1")}
    assert not [f for f in os.listdir(str(tmp_path)) if ".part" in f or ".tmp" in f]
    # (a second call finds the table complete and returns, like the reference's extract_embedding.py:126-128)
    again = subprocess.run([sys.executable, str(script)] + flags("three"), env=dict(base_env, XVECTOR_SHARD_OUTPUT="files"),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert again.returncode == 0 and "Both output ark and scp files exist" in again.stdout.decode()
