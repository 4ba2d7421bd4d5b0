"""The reference's OWN caller scripts driving this build's twins (CPU only, runs only where /root/reference is mounted).

The drop-in claim of SURVEY §8b is that ``local/tf/models.py`` + ``kaldi_io.py`` can be swapped under the reference's
unchanged callers.  Here the reference's ``train_dnn_one_iteration.py`` (with its ``examples_io.TarFileDataLoader``) and
``extract_embedding.py`` are imported from /root/reference and run against this build's ``models`` / ``kaldi_io`` modules;
the GPU work is replaced by recording fakes (training) or the CPU oracle (extraction), so what is pinned is the
interface: constructor, method names, argument order, ``args`` fields, the data-loader protocol, stream types, the
model-directory contract and the log lines the reference's own parser (ze_utils.get_successful_models) reads.
Nothing from the reference is copied: its modules are imported in place, under the py2->py3 shims SURVEY §8c lists.
"""
import importlib.util
import io
import logging
import os
import sys
import tarfile
import types

import numpy as np
import pytest

REF = "/root/reference/local/tf"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")


def _load_ref(name, shims):
    """Import /root/reference/local/tf/<name>.py as module ``name`` with ``shims`` pre-installed in sys.modules."""
    for k, v in shims.items():
        sys.modules[k] = v
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def ref_env(monkeypatch):
    """sys.modules arranged as in a reference checkout whose models.py / kaldi_io.py were replaced by this build's."""
    import _thread
    import kaldi_io      # this build's (tests/conftest.py put local/tf on sys.path)
    import models        # this build's
    saved = {k: sys.modules.get(k) for k in ("ze_utils", "examples_io", "train_dnn_one_iteration", "extract_embedding", "thread",
                                             "h5py", "mkl", "numexpr")}
    threads = types.ModuleType("mkl"); threads.set_num_threads = lambda n: None
    nx = types.ModuleType("numexpr"); nx.set_num_threads = lambda n: None
    shims = {"thread": _thread, "h5py": types.ModuleType("h5py"), "mkl": threads, "numexpr": nx}
    # numpy 2.x probes f.fileno() on the tar member the reference's TarFileDataLoader hands to np.load
    # (examples_io.py:245); python 3.10's tarfile._FileInFile has none -> make it answer like any non-file stream
    def _no_fileno(self):
        raise io.UnsupportedOperation("fileno")
    monkeypatch.setattr(tarfile._FileInFile, "fileno", _no_fileno, raising=False)
    ze = _load_ref("ze_utils", shims)                       # the reference's full ze_utils (py2 'import thread')
    ex = _load_ref("examples_io", {})
    assert sys.modules["models"] is models and sys.modules["kaldi_io"] is kaldi_io
    yield dict(ze_utils=ze, examples_io=ex, models=models, kaldi_io=kaldi_io)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


class _FakeTrainer(object):
    def __init__(self, weights):
        self.weights, self.calls = weights, []

    def step(self, x, labels, lr, dropout_proportion=0.0, seed=0):
        self.calls.append((x.shape, x.dtype, np.asarray(labels).copy(), lr, dropout_proportion, seed))
        return 2.0 - 0.1 * len(self.calls), 0.25 * len(self.calls)

    def export(self):
        return self.weights, dict(t=len(self.calls), m={}, v={})


def test_reference_train_dnn_one_iteration_drives_the_twin(ref_env, tmp_path, monkeypatch, capsys):
    models = ref_env["models"]
    from xvector_amd import weights as wio
    in_dir, out_dir = str(tmp_path / "model_0"), str(tmp_path / "model_1")
    models.ModelWithoutDropout().build_model(8, 5, in_dir)                       # what train_dnn.py --stage -1 does
    assert ref_env["ze_utils"].is_correct_model_dir(in_dir)                      # the reference's own predicate accepts it
    # an egs archive in the reference's format (examples_io.save_data_info_tar): minibatch_<i>.npy float16 + <tar>.npy labels
    rng = np.random.default_rng(0)
    tar_path = str(tmp_path / "egs.1.tar")
    mats = [rng.standard_normal((4, 30 + 5 * i, 5)).astype(np.float16) for i in range(3)]
    labels = rng.integers(0, 8, (3, 4)).astype(np.int32)
    with tarfile.TarFile(tar_path, "w") as tf:
        for i, m in enumerate(mats):
            buf = io.BytesIO(); np.save(buf, m); size = buf.tell(); buf.seek(0)
            info = tarfile.TarInfo(name="minibatch_%d.npy" % i); info.size = size
            tf.addfile(tarinfo=info, fileobj=buf)
    np.save(tar_path.replace(".tar", ".npy"), labels)
    fake = {}

    def fake_trainer(self, input_dir, logger, first_batch=None):
        w, meta = wio.load_model_dir(input_dir)
        self.meta, self.num_classes = meta, meta["num_classes"]
        fake["tr"] = _FakeTrainer(w)
        return fake["tr"]
    monkeypatch.setattr(models.Model, "_trainer", fake_trainer)
    monkeypatch.setattr(sys, "argv", ["train_dnn_one_iteration.py", "--feature-dim", "5", "--minibatch-size", "4", "--minibatch-count", "3",
                                      "--learning-rate", "0.00125", "--dropout-proportion", "0.1", "--random-seed", "7", "--print-interval",
                                      "2", "--tar-file", tar_path, "--input-dir", in_dir, "--output-dir", out_dir])
    ref_train = _load_ref("train_dnn_one_iteration", {})
    ref_train.train(ref_train.get_args())
    tr = fake["tr"]
    assert len(tr.calls) == 3
    for (shape, dtype, lab, lr, dp, seed), m, l in zip(tr.calls, mats, labels):
        assert shape == m.shape and dtype == np.float16 and np.array_equal(lab, l)
        assert lr == 0.00125 and dp == 0.1 and seed == 7
    assert ref_env["ze_utils"].is_correct_model_dir(out_dir)
    assert wio.load_optimizer_state(out_dir)["t"] == 3
    # the log lines are what the reference's own parser reads back (ze_utils.py:123-154)
    log = tmp_path / "train.1.log"
    log.write_text(capsys.readouterr().out)
    text = log.read_text()
    assert "Average training loss for minibatches 1-2 is" in text and "Overall average objective function is" in text
    accepted, best = ref_env["ze_utils"].get_successful_models(1, str(tmp_path / "train.%.log"))
    assert accepted == [1] and best == 1


def test_eval_logs_parse_with_the_reference_accuracy_report(ref_env, tmp_path, monkeypatch):
    """Model.eval's log (written the way eval_dnn.py does, one file per iteration) is read back by the reference's own
    ze_utils.parse_prob_logs (ze_utils.py:490-528, the source of accuracy.report)."""
    models = ref_env["models"]

    class FakeEval(object):
        def __init__(self, loss, acc):
            self.v = (loss, acc)

        def eval_batch(self, x, labels):
            return self.v

    class OneBatch(object):
        count = 2

        def __init__(self):
            self.left = 2

        def pop(self, timeout=30):
            self.left -= 1
            return (np.zeros((3, 20, 5), np.float16), np.zeros(3, np.int32)) if self.left >= 0 else (None, None)

    (tmp_path / "log").mkdir()
    want = {}
    for it, (tl, ta, vl, va) in {3: (0.6923, 0.8548, 0.9011, 0.7712), 7: (0.4102, 0.9017, 0.8125, 0.8003)}.items():
        for kind, loss, acc in (("train_subset", tl, ta), ("valid", vl, va)):
            path = tmp_path / "log" / ("compute_prob_%s.%d.log" % (kind, it))
            logger = logging.getLogger("eval_%s_%d" % (kind, it))
            logger.setLevel(logging.INFO)
            h = logging.FileHandler(str(path), mode="w")
            h.setFormatter(logging.Formatter("%(asctime)s [%(pathname)s:%(lineno)s - %(funcName)s - %(levelname)s ] %(message)s"))
            logger.addHandler(h)
            monkeypatch.setattr(models.Model, "_trainer", lambda self, d, lg, _l=loss, _a=acc: FakeEval(_l, _a))
            models.Model().eval(OneBatch(), "unused", True, logger)
            h.close()
            logger.removeHandler(h)
        want[it] = (tl, ta, vl, va)
    rows = ref_env["ze_utils"].parse_prob_logs(str(tmp_path), key="accuracy")
    assert [r[0] for r in rows] == [3, 7]
    for it, tl, ta, vl, va in rows:
        assert (tl, ta, vl, va) == pytest.approx(want[it], abs=1e-4)
    # the driver's per-job logs end with the trailer queue.pl/run.pl leave ("# Accounting: time=<s> threads=1"): the
    # reference's generate_report (ze_utils.py:467-558) and the twin's produce the same accuracy.report from this directory
    for it, job, sec in ((3, 1, 41), (3, 2, 57), (7, 1, 12)):
        (tmp_path / "log" / ("train.%d.%d.log" % (it, job))).write_text(
            "2026-01-01 00:00:00,000 [x.py:1 - f - INFO ] Overall average objective function is -0.5 over 9 segments.\n"
            "# Accounting: time=%d threads=1\n" % sec)
    from conftest import TWIN                   # sys.modules["ze_utils"] is the reference's here: load the twin by path
    spec = importlib.util.spec_from_file_location("ze_utils_twin", os.path.join(TWIN, "ze_utils.py"))
    twin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(twin)
    assert twin.__file__ != ref_env["ze_utils"].__file__
    ref_report, ref_times, ref_data = ref_env["ze_utils"].generate_report(str(tmp_path))
    report, times, data = twin.generate_report(str(tmp_path))
    assert times == ref_times == {3: 57.0, 7: 12.0}
    assert data == ref_data and report == ref_report


def test_egs_archives_interchange_with_the_reference_loader(ref_env, tmp_path):
    """An archive written by this build's examples_io.write_egs_tar is read by the reference's TarFileDataLoader
    (examples_io.py:223-255) exactly as by this build's loader."""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "x-vector-kaldi-tf_amd", "local", "tf")
    spec = importlib.util.spec_from_file_location("twin_examples_io", os.path.join(here, "examples_io.py"))
    twin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(twin)
    rng = np.random.default_rng(3)
    mats = [rng.standard_normal((3, 25 + i, 4)).astype(np.float32) for i in range(4)]
    labels = rng.integers(0, 5, (4, 3)).astype(np.int32)
    tar = str(tmp_path / "egs.1.tar")
    twin.write_egs_tar(tar, mats, labels)
    ref_loader = ref_env["examples_io"].TarFileDataLoader(tar, queue_size=8)
    mine = twin.TarFileDataLoader(tar, queue_size=8)
    assert ref_loader.count == mine.count == 4
    for _ in range(4):
        (a, la), (b, lb) = ref_loader.pop(timeout=10), mine.pop(timeout=10)
        assert a.dtype == b.dtype == np.float16 and np.array_equal(a, b) and np.array_equal(la, lb)


def test_reference_extract_embedding_cli_drives_the_twin(ref_env, tmp_path, monkeypatch):
    """Reference extract_embedding.py (its own argparse, open_or_fd calls and stream handling) -> this build's
    Model.make_embedding; the GPU extractor is replaced by the CPU oracle, so the output ark must hold the oracle's
    x-vectors for exactly the keys the reference driver would keep."""
    models, kaldi_io = ref_env["models"], ref_env["kaldi_io"]
    from oracle import oracle
    from xvector_amd import engine, synthetic
    topo = synthetic.SMALL_TOPOLOGY
    w = synthetic.trained_like(topo, 5, num_classes=8, seed=11)
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=8, feat_dim=5), mdir, None)
    rng = np.random.default_rng(0)
    utts = [("u%02d" % i, (rng.standard_normal((T, 5)) * 3).astype(np.float32)) for i, T in enumerate([30, 10, 250, 0, 99])]
    feats = str(tmp_path / "feats.ark")
    with open(feats, "wb") as f:
        for k, m in utts:
            kaldi_io.write_mat(f, m, key=k)

    class FakeExtractor(object):
        def __init__(self, model, min_chunk_size, chunk_size, **kw):
            self.a = (min_chunk_size, chunk_size)
            self.stats = dict(batches=0, chunks=0, frames=0, rows=0)

        def extract(self, mats):
            self.stats["frames"] += sum(m.shape[0] for m in mats)
            return [oracle.embed_utterance(m, w, topo, self.a[0], self.a[1], np.float32) for m in mats]

        def submit(self, mats, addrs=None):        # the extractor's two-phase interface (launch now, collect later)
            assert addrs is None or len(addrs) == len(mats)
            return self.extract(mats)

        def finish(self, handle, as_array=False):
            if not as_array:
                return handle
            valid = np.array([v is not None for v in handle], dtype=bool)
            full = np.zeros((len(handle), topo["embedding_sizes"][0]), np.float32)
            for i, v in enumerate(handle):
                if v is not None:
                    full[i] = v
            return full, valid

    def fake_load(self, sess, input_dir, logger):
        self.meta = dict(topology=topo)
        self.device_model = types.SimpleNamespace(device="cpu", embed_dim=topo["embedding_sizes"][0], feat_dim=23)
    monkeypatch.setattr(engine, "Extractor", FakeExtractor)
    monkeypatch.setattr(models.Model, "load_model", fake_load)
    out = str(tmp_path / "xvector.ark")
    monkeypatch.setattr(sys, "argv", ["extract_embedding.py", "--use-gpu", "no", "--min-chunk-size", "25", "--chunk-size", "100",
                                      "--feature-rspecifier", "ark:" + feats, "--vector-wspecifier", "| cat > " + out, "--model-dir", mdir])
    # (the reference always writes through a pipe -- '| copy-vector ark:- ark,scp:...' -- because it opens the wspecifier
    #  with open_or_fd's default mode; a plain 'ark:file' wspecifier does not work in the reference either)
    ref_cli = _load_ref("extract_embedding", {})
    ref_cli.eval_dnn(ref_cli.get_args())
    import time
    for _ in range(100):              # the pipe's subprocess is reaped by a helper thread (kaldi_io.popen): wait for `cat`
        if os.path.exists(out) and os.path.getsize(out) >= 3 * (4 + 15 + 4 * topo["embedding_sizes"][0]):
            break
        time.sleep(0.05)
    got = list(kaldi_io.read_vec_flt_ark(out))
    assert [k for k, _ in got] == ["u00", "u02", "u04"]            # T=10 (< min chunk) and T=0 are skipped, order kept
    for k, v in got:
        ref = oracle.embed_utterance(dict(utts)[k], w, topo, 25, 100, np.float32)
        assert v.dtype == np.float32 and np.array_equal(v, ref)
