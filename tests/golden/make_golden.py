#!/usr/bin/env python
"""Generate the committed golden fixtures in tests/golden/.

Runs ONLY in the build container (it imports the reference's Python from /root/reference, which
never travels to the GPU box).  Nothing here is copied from the reference: the fixtures are DATA --
bytes the reference's own ``kaldi_io`` writes / arrays it decodes, and the output stream the
reference's own ``Model.make_embedding`` (local/tf/models.py:356-432) produces, and -- since round 6 -- the values the
reference's own GRAPH CODE computes: every ``build_model`` of local/tf/models.py (+ tf_block.py) is executed under
``tests/golden/numpy_tf1.py`` (a NumPy float64 evaluator registered as ``tensorflow``; TensorFlow itself is absent), the
fixture's weights are written into the checkpoint that ``build_model`` saved BY THE VARIABLE NAMES ITS SCOPES PRODUCED, and
the reference's ``load_model`` / ``make_embedding`` / ``train_one_iteration`` / ``eval`` run on it unmodified.

Fixtures
  ark_io.npz            bytes written by reference write_mat/write_vec_flt (FM, DM, FV, DV), a
                        hand-built 'CM ' record and an ascii record + what reference read_mat decodes
  make_embedding.npz    for (min_chunk, chunk) settings: utterance lengths, the chunk lengths each
                        key was run with, and the exact output ark bytes of the reference driver
  schedules.npz         reference ze_utils.get_learning_rate / get_dropout_edit_string (ze_utils.py:111-120, 310-443)
                        evaluated on grids of arguments (SURVEY §8c golden 4)
  egs_ranges.npz        what the reference's ranges/scp loader (examples_io.py:12-75,188-221) serves for a toy table
  forward_refgraph.npz  x-vectors (embedding[0] and [1]), pooled vectors and sub-sampled per-layer tensors that the reference's
                        graphs (all 8 classes, eval phase) produce for trained-like weights from a seed, T in {25,200,400,1000};
                        the variable name / shape table and the initial-value law of each build_model; the output ark of the
                        reference's make_embedding driving its own graph (every value asserted <= 1e-12 from the fp64 oracle)
  train_refgraph.npz    the reference's train_one_iteration (3 Adam steps, lr 1e-4; class Model with dropout 0.2 and the
                        masks drawn) and eval on every class: per-step loss / accuracy, step-0 gradients, weights, moving
                        statistics and Adam slots after the steps (small tensors whole, large ones strided)
Usage:  python tests/golden/make_golden.py
"""
import io
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/local/tf"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))

from oracle import oracle  # noqa: E402
from xvector_amd import synthetic, topology  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fixture_inputs import (CONTROL_FEAT, CONTROL_LENGTHS, CONTROL_SEED, CONTROL_SETTINGS, FWD_SEED, FWD_T,  # noqa: E402
                            control_inputs)


def control_stub():
    """The tensorflow stand-in of the CONTROL-FLOW fixture (make_embedding.npz): Session.run hands the chunk to a callback.  Lengths
    up to 30010 frames x 5 settings make the full-width graphs too slow for that fixture; it pins which chunks are run and the bytes
    written, not arithmetic (the reference's graph arithmetic is pinned by forward_refgraph.npz)."""
    tf = types.ModuleType("tensorflow")

    class _Tensor(object):
        def __init__(self, name, shape=None):
            self.name, self.shape = name, shape

    class _Graph(object):
        def __init__(self, num_classes):
            self.num_classes = num_classes

        def get_tensor_by_name(self, name):
            return _Tensor(name, (None, self.num_classes) if name == "input_y:0" else None)

        def get_operation_by_name(self, name):
            return _Tensor(name)

    class Session(object):
        forward = None          # set by the driver: f(mat[T,F]) -> float64[E]
        log = None

        def __init__(self, config=None, graph=None):
            self.graph = _Graph(8)

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def run(self, fetch, feed_dict=None):
            assert fetch.name == "embed_layer-0/scores:0"          # models.py:159,414
            x = [v for k, v in feed_dict.items() if getattr(k, "name", "") == "input_x:0"][0]
            assert x.ndim == 3 and x.shape[0] == 1                  # batch 1, models.py:410
            Session.log.append(int(x.shape[1]))
            return Session.forward(x[0]).astype(np.float32)[None, :]

    class _Saver(object):
        def restore(self, sess, path):
            pass

    tf.Session = Session
    tf.ConfigProto = lambda **kw: types.SimpleNamespace()
    tf.train = types.SimpleNamespace(import_meta_graph=lambda path: _Saver())
    return tf, Session


def import_reference():
    """Import the reference's kaldi_io / models with the minimum of sys.modules shims: ``tensorflow`` is
    tests/golden/numpy_tf1.py, ``thread`` is python 3's _thread (ze_utils.py:10)."""
    import _thread
    sys.modules.setdefault("thread", _thread)          # ze_utils.py:10 is python-2 'import thread'
    import numpy_tf1
    numpy_tf1.install()
    sys.path.insert(0, REF)
    import kaldi_io as ref_kaldi_io
    import models as ref_models
    sys.path.remove(REF)
    ref_models.set_cuda_visible_devices = lambda **kw: None        # ze_utils.py:25-46 shells out to nvidia-smi; not on the path
    # keep our own modules importable under their names afterwards
    for m in ("kaldi_io", "models", "ze_utils", "tf_block"):
        sys.modules["ref_" + m] = sys.modules.pop(m)
    ctl_tf, Session = control_stub()
    ref_models.ctl_tf = ctl_tf
    return ref_kaldi_io, ref_models, Session


class _Log(object):
    def info(self, *a):
        pass

    def warning(self, *a):
        pass


def golden_ark_io(ref_io, out):
    rng = np.random.default_rng(7)
    fm = rng.standard_normal((6, 23)).astype(np.float32)
    dm = rng.standard_normal((4, 5))
    fv = rng.standard_normal(512).astype(np.float32)
    dv = rng.standard_normal(7)
    bio = io.BytesIO(); bio.mode = "wb"
    ref_io.write_mat(bio, fm, key="utt-a.1")
    ref_io.write_mat(bio, dm, key="utt_b/2")
    ref_io.write_mat(bio, np.zeros((0, 23), np.float32), key="empty")
    mat_ark = bio.getvalue()
    bio = io.BytesIO(); bio.mode = "wb"
    ref_io.write_vec_flt(bio, fv, key="spk1")
    ref_io.write_vec_flt(bio, dv, key="spk2")
    vec_ark = bio.getvalue()
    # hand-built 'CM ' record (Kaldi CompressedMatrix format 1): global header, per-column
    # percentile headers (uint16), column-major uint8 payload
    rows, cols = 9, 4
    hdr = np.array([(-3.5, 7.25)], dtype="<f4").tobytes() + np.array([rows, cols], "<i4").tobytes()
    pct = np.sort(rng.integers(0, 65536, size=(cols, 4)), axis=1).astype("<u2")
    data = rng.integers(0, 256, size=(cols, rows)).astype(np.uint8)
    data[0, :4] = [0, 64, 65, 192]
    data[1, :3] = [193, 255, 128]
    cm_ark = b"cm1 " + b"\x00BCM " + hdr + pct.tobytes() + data.tobytes()
    cm_dec = [m for _, m in ref_io.read_mat_ark(io.BytesIO(cm_ark))][0]
    txt_ark = b"txt1  [\n  1.5 -2 3e-1\n  4 5 6.25 ]\n"
    txt_dec = [m for _, m in ref_io.read_mat_ark(io.BytesIO(txt_ark))][0]
    vec_txt = b"vtx  [ 1 2.5 -3 ]\n"
    vec_txt_dec = [v for _, v in ref_io.read_vec_flt_ark(io.BytesIO(vec_txt))][0]
    np.savez_compressed(out, fm=fm, dm=dm, fv=fv, dv=dv,
                        mat_ark=np.frombuffer(mat_ark, np.uint8), vec_ark=np.frombuffer(vec_ark, np.uint8),
                        cm_ark=np.frombuffer(cm_ark, np.uint8), cm_dec=np.asarray(cm_dec),
                        txt_ark=np.frombuffer(txt_ark, np.uint8), txt_dec=np.asarray(txt_dec),
                        vec_txt=np.frombuffer(vec_txt, np.uint8), vec_txt_dec=np.asarray(vec_txt_dec))
    print("ark_io: mat_ark %d B, vec_ark %d B, cm %s" % (len(mat_ark), len(vec_ark), cm_dec.shape))


def golden_make_embedding(ref_io, ref_models, Session, out):
    topo = synthetic.SMALL_TOPOLOGY
    weights = synthetic.trained_like(topo, CONTROL_FEAT, num_classes=8, seed=CONTROL_SEED)
    utts = control_inputs()
    bio = io.BytesIO(); bio.mode = "wb"
    for k, m in utts:
        ref_io.write_mat(bio, m, key=k)
    in_ark = bio.getvalue()
    Session.forward = staticmethod(lambda x: oracle.forward(x, weights, topo, np.float64))
    res = {}
    graph_tf, ref_models.tf = ref_models.tf, ref_models.ctl_tf
    for si, (min_chunk, chunk) in enumerate(CONTROL_SETTINGS):
        Session.log = []
        so = io.BytesIO(); so.mode = "wb"
        ref_models.Model().make_embedding(io.BytesIO(in_ark), so, "/nonexistent", min_chunk, chunk, False, _Log())
        res["out_ark_%d" % si] = np.frombuffer(so.getvalue(), np.uint8)
        res["chunk_lens_%d" % si] = np.array(Session.log, np.int64)
        print("make_embedding[min=%d chunk=%d]: %d chunks run, %d B out" %
              (min_chunk, chunk, len(Session.log), len(so.getvalue())))
    ref_models.tf = graph_tf
    np.savez_compressed(out, settings=np.array(CONTROL_SETTINGS, np.int64),
                        lengths=np.array(CONTROL_LENGTHS, np.int64), seed=CONTROL_SEED, feat=CONTROL_FEAT, **res)


REF_CLASSES = [    # (fixture tag, class in the reference's local/tf/models.py, T list)
    ("default", "ModelWithoutDropout", FWD_T), ("dilated", "ModelWithoutDropoutTdnn", FWD_T),
    ("prelu", "ModelWithoutDropoutPRelu", [25, 200]), ("lrelu", "ModelL2LossWithoutDropoutLRelu", [25, 200]),
    ("attention", "ModelL2LossWithoutDropoutLReluAttention", [25, 200, 1000]),
    ("dropout", "Model", [25, 200]), ("l2prelu", "ModelL2LossWithoutDropoutPRelu", [25, 200]),
    ("heinit", "ModelL2LossWithoutDropoutReluHeInit", [25, 200])]
REFGRAPH_EMBED_LENGTHS = [0, 24, 25, 99, 100, 200, 450, 1000, 1010]
REFGRAPH_EMBED_SETTINGS = [(25, 300), (100, -1), (25, -1)]


def _build_with_reference(ref_models, cls, num_classes, feat_dim, weights=None):
    """model dir the reference's own <cls>.build_model wrote (under numpy_tf1); optionally the fixture's weights stored into its
    checkpoint by variable name.  -> (dir, {name: initial value})"""
    import tempfile
    import numpy_tf1
    d = tempfile.mkdtemp(prefix="refgraph_")
    getattr(ref_models, cls)().build_model(num_classes, feat_dim, d, _Log())
    init = numpy_tf1.read_checkpoint(os.path.join(d, "model"))
    if weights is not None:
        mine = sorted(k for k in init if "/Adam" not in k and not k.startswith("beta"))
        assert mine == sorted(weights), (cls, sorted(set(mine) ^ set(weights)))      # the names the reference's scopes produce
        numpy_tf1.write_checkpoint_values(os.path.join(d, "model"), weights)
    return d, init


def golden_forward_refgraph(ref_io, ref_models, out):
    import shutil
    import numpy_tf1 as tf
    res = {}
    for tname, cls, Ts in REF_CLASSES:
        topo = topology.get(cls)
        weights = synthetic.trained_like(topo, 23, seed=FWD_SEED)
        mdir, init = _build_with_reference(ref_models, cls, 64, 23, weights)
        names = sorted(init)
        res["%s_var_names" % tname] = np.array(names)
        res["%s_var_shapes" % tname] = np.array([",".join(str(d) for d in init[k].shape) for k in names])
        # the law of build_model's initial values (models.py:56-58,82-84,98-100; tf_block.py:10-14,45-46; He variants 1158-1210):
        # per variable [min, max, mean, std]
        res["%s_init_stats" % tname] = np.array([[init[k].min(), init[k].max(), init[k].mean(), init[k].std()] for k in names])
        rng = np.random.default_rng(FWD_SEED + 1)
        tf.reset_default_graph()
        tf.FETCH_FLOAT64[0] = True
        with tf.Session() as sess:
            m = getattr(ref_models, cls)()
            m.load_model(sess, mdir, _Log())                 # the reference picks the tensors (models.py:143-162)
            g = sess.graph
            pooled_t = g.get_tensor_by_name("embed_layer-0/scores/MatMul:0").inputs[0]
            layer_t = [g.get_tensor_by_name("frame_level_info_layer-%d/cond/Merge:0" % i) for i in range(5)]
            for T in Ts:
                x = (rng.standard_normal((T, 23)) * 3.0).astype(np.float32)
                feed = {m.input_x: x[None], m.dropout_keep_prob: 1.0, m.phase: False}         # models.py:412
                e0, e1, pooled = sess.run([m.embedding[0], m.embedding[1], pooled_t], feed)
                layers = sess.run(layer_t, feed) if T == 25 else None
                o1, inter = oracle.forward(x, weights, topo, np.float64, embedding_index=1, return_intermediates=True)
                assert e0.dtype == np.float64 and e0.shape == (1, 512)
                for a, b, what in ((e0[0], inter[6], "e0"), (e1[0], o1, "e1"), (pooled[0], inter[5], "pooled")):
                    assert oracle.rel_l2(a, b) < 1e-12, (cls, T, what, oracle.rel_l2(a, b))
                res["%s_T%d_e0" % (tname, T)] = e0[0]
                res["%s_T%d_e1" % (tname, T)] = e1[0]
                if T == 25:
                    for li in range(5):
                        assert oracle.rel_l2(layers[li][0], inter[li]) < 1e-12, (cls, li)
                        res["%s_T25_layer%d_sub" % (tname, li)] = layers[li][0][:, ::16].copy()
                    res["%s_T25_pooled" % tname] = pooled[0]
                print("refgraph forward %s (%s) T=%d |e0|=%.4f  oracle-refgraph %.1e" %
                      (tname, cls, T, np.linalg.norm(e0), oracle.rel_l2(e0[0], inter[6])))
        tf.FETCH_FLOAT64[0] = False
        if tname == "default":
            # the reference's make_embedding driving the reference's graph, ark bytes in -> ark bytes out (float32 fetches, as TF's)
            rng = np.random.default_rng(FWD_SEED + 2)
            bio = io.BytesIO(); bio.mode = "wb"
            for i, T in enumerate(REFGRAPH_EMBED_LENGTHS):
                ref_io.write_mat(bio, (rng.standard_normal((T, 23)) * 3.0).astype(np.float32), key="rg%02d-T%d" % (i, T))
            for si, (min_chunk, chunk) in enumerate(REFGRAPH_EMBED_SETTINGS):
                so = io.BytesIO(); so.mode = "wb"
                tf.reset_default_graph()
                ref_models.ModelWithoutDropout().make_embedding(io.BytesIO(bio.getvalue()), so, mdir, min_chunk, chunk, False, _Log())
                res["embed_out_ark_%d" % si] = np.frombuffer(so.getvalue(), np.uint8)
                print("refgraph make_embedding[min=%d chunk=%d]: %d B out" % (min_chunk, chunk, len(so.getvalue())))
            res["embed_lengths"] = np.array(REFGRAPH_EMBED_LENGTHS, np.int64)
            res["embed_settings"] = np.array(REFGRAPH_EMBED_SETTINGS, np.int64)
        shutil.rmtree(mdir)
    np.savez_compressed(out, seed=FWD_SEED, **res)


TRAIN_CLASSES = ["ModelWithoutDropout", "ModelWithoutDropoutTdnn", "ModelWithoutDropoutPRelu", "ModelL2LossWithoutDropoutPRelu",
                 "ModelL2LossWithoutDropoutLRelu", "ModelL2LossWithoutDropoutLReluAttention", "ModelL2LossWithoutDropoutReluHeInit", "Model"]
TRAIN_SEED, TRAIN_CLASSES_N, TRAIN_LR, TRAIN_STRIDE = 77, 7, 0.0001, 1999


class _Batches(object):
    """The data_loader duck type of examples_io.py:213-221 (count, pop)."""

    def __init__(self, batches):
        self.batches, self.count = list(batches), len(batches)

    def pop(self, timeout=30):
        return self.batches.pop(0) if self.batches else (None, None)


def _compact(a, small_stride=1):
    """Tensors of up to 1536 elements whole (or every small_stride-th element), larger ones every TRAIN_STRIDE-th element (flat)."""
    a = np.asarray(a, np.float64).reshape(-1)
    return a[::small_stride].copy() if a.size <= 1536 else a[::TRAIN_STRIDE].copy()


def golden_train_refgraph(ref_models, out):
    """Reference train_one_iteration (models.py:216-305) + eval (models.py:307-354) on every class, executed under numpy_tf1."""
    import shutil
    import tempfile
    import numpy_tf1 as tf
    res = {}
    for cls in TRAIN_CLASSES:
        topo = topology.get(cls)
        weights = synthetic.trained_like(topo, 23, num_classes=TRAIN_CLASSES_N, seed=TRAIN_SEED)
        mdir, _ = _build_with_reference(ref_models, cls, TRAIN_CLASSES_N, 23, weights)
        odir = tempfile.mkdtemp(prefix="refgraph_out_")
        rng = np.random.default_rng(TRAIN_SEED + 1)
        batches = [((rng.standard_normal((6, 40 + 5 * i, 23)) * 3).astype(np.float16), rng.integers(0, TRAIN_CLASSES_N, 6).astype(np.int32))
                   for i in range(3)]                                        # fp16 minibatches, one length each (examples_io.py:165,176)
        steps = []

        def hook(fetches, run):
            if isinstance(fetches, list) and len(fetches) == 3:              # [optimizer, loss, accuracy] (models.py:262)
                steps.append((float(run.val[id(fetches[1])]), float(run.val[id(fetches[2])]), run.aux.get("gradients")))
            elif isinstance(fetches, list) and len(fetches) == 2:            # [loss, accuracy] (models.py:340)
                steps.append((float(run.val[id(fetches[0])]), float(run.val[id(fetches[1])]), None))
        tf.RUN_HOOK[0] = hook
        tf.DROPOUT_LOG[:] = []
        args = types.SimpleNamespace(learning_rate=TRAIN_LR, print_interval=10, dropout_proportion=0.2 if cls == "Model" else 0.0,
                                     input_dir=mdir, output_dir=odir, random_seed=5)
        tf.reset_default_graph()
        getattr(ref_models, cls)().train_one_iteration(_Batches(batches), args, _Log())
        after = tf.read_checkpoint(os.path.join(odir, "model"))
        train_steps, steps[:] = list(steps), []
        tf.reset_default_graph()
        getattr(ref_models, cls)().eval(_Batches(batches[:2]), odir, False, _Log())
        eval_steps = list(steps)
        tf.RUN_HOOK[0] = None
        assert len(train_steps) == 3 and len(eval_steps) == 2
        res["%s/loss" % cls] = np.array([s[0] for s in train_steps])
        res["%s/accuracy" % cls] = np.array([s[1] for s in train_steps])
        res["%s/eval_loss" % cls] = np.array([s[0] for s in eval_steps])
        res["%s/eval_accuracy" % cls] = np.array([s[1] for s in eval_steps])
        for k, g in train_steps[0][2].items():
            res["%s/grad0/%s" % (cls, k)] = _compact(g)
        for k, v in after.items():
            res["%s/after/%s" % (cls, k)] = _compact(v, 8 if "/Adam" in k else 1)
        if cls == "Model":                                                   # dropout sites and masks (models.py:70-72,92-94)
            assert len(tf.DROPOUT_LOG) == 15 + 10, len(tf.DROPOUT_LOG)       # 5 sites x 3 training steps, then 5 x 2 eval steps at keep 1.0
            for n, (name, mask, keep) in enumerate(tf.DROPOUT_LOG[:15]):
                res["Model/dropout/%d/%d/%s" % (n // 5, n % 5, name.replace("/", "|"))] = np.packbits(mask.astype(np.uint8).reshape(-1))
                assert keep == 0.8
            assert all(keep == 1.0 and mask.all() for _, mask, keep in tf.DROPOUT_LOG[15:])
        # agreement with the training oracle, asserted at generation time too
        from oracle import train_ref
        ww = {k: np.asarray(v, np.float64) for k, v in weights.items()}
        adam = {"t": 0, "m": {}, "v": {}}
        for bi, (x, l) in enumerate(batches):
            dr = None
            if cls == "Model":
                dr = dict((name.split("/")[0], (mask, keep)) for name, mask, keep in tf.DROPOUT_LOG[bi * 5:(bi + 1) * 5])
            loss, acc, ww, adam, grads = train_ref.train_step(ww, adam, topo, x.astype(np.float64), l, TRAIN_LR, dropout=dr)
            assert abs(loss - train_steps[bi][0]) < 1e-10 * max(1.0, abs(loss)), (cls, bi, loss, train_steps[bi][0])
        worst = max(oracle.rel_l2(after[k], ww[k]) for k in ww)
        assert worst < 1e-9, (cls, worst)
        print("refgraph train %s: losses %s, eval %s, oracle-refgraph after 3 steps %.1e" %
              (cls, ["%.4f" % s[0] for s in train_steps], ["%.4f" % s[0] for s in eval_steps], worst))
        tf.DROPOUT_LOG[:] = []
        shutil.rmtree(mdir); shutil.rmtree(odir)
    rng = np.random.default_rng(TRAIN_SEED + 1)
    np.savez_compressed(out, seed=TRAIN_SEED, num_classes=TRAIN_CLASSES_N, lr=TRAIN_LR, stride=TRAIN_STRIDE, **res)


def golden_schedules(out):
    ref = sys.modules["ref_ze_utils"]
    lr_args, lr_vals = [], []
    for num_iters in (1, 7, 120):
        for num_jobs in (1, 3, 8):
            total = 4 * num_iters + 3
            for it in sorted(set([0, 1, num_iters // 2, max(num_iters - 2, 0), num_iters - 1])):
                for done in (0, it * 2, total - 1, total):
                    for ini, fin in ((0.001, 0.0001), (3e-4, 3e-4), (1e-3, 2e-5)):
                        lr_args.append((it, num_jobs, num_iters, done, total, ini, fin))
                        lr_vals.append(ref.get_learning_rate(it, num_jobs, num_iters, done, total, ini, fin))
    scheds = ["0,0@0.10,0.1@0.50,0", "0.0,0.3,0.0", "0.2,0.2", "0,0.5@0.25,0.5@0.25,0.1@0.8,0", "0.1,0@0.0,0.4@1.0,0.2"]
    fracs = [0.0, 0.05, 0.1, 0.25, 0.3, 0.5, 0.75, 0.8, 0.999, 1.0]
    table = np.array([[ref.get_dropout_edit_string(sc, f) for f in fracs] for sc in scheds], dtype=np.float64)
    bad = ["0.1", "0,0.1@0.7,0.2@0.3,0", "0,0.2@1.5,0", "0,x@0.5,0"]
    bad_raises = []
    for b in bad:
        try:
            ref.get_dropout_edit_string(b, 0.5)
            bad_raises.append(0)
        except Exception:
            bad_raises.append(1)
    assert ref.get_dropout_edit_string(None, 0.3) is None
    np.savez_compressed(out, lr_args=np.array(lr_args, np.float64), lr_vals=np.array(lr_vals, np.float64),
                        schedules=np.array(scheds), fractions=np.array(fracs), dropout_table=table,
                        bad_schedules=np.array(bad), bad_raises=np.array(bad_raises))
    print("schedules: %d learning rates, %dx%d dropout table, bad strings raise: %s" %
          (len(lr_vals), table.shape[0], table.shape[1], bad_raises))


def golden_egs_ranges(out):
    """Reference examples_io.process_range_file + load_ranges_data + DataLoader (the ranges/scp input mode of
    train_dnn_one_iteration.py:177-200): the minibatches it serves, in the order it serves them, for a toy feature
    table and ranges file, with --shuffle and a seed as train() applies them."""
    import tempfile
    h5 = types.ModuleType("h5py")
    sys.modules.setdefault("h5py", h5)
    sys.path.insert(0, REF)
    sys.modules["kaldi_io"] = sys.modules["ref_kaldi_io"]
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_examples_io", os.path.join(REF, "examples_io.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    sys.path.remove(REF)
    del sys.modules["kaldi_io"]
    ref_io = sys.modules["ref_kaldi_io"]
    rng = np.random.default_rng(42)
    F, B, count = 6, 4, 5
    utts = [("utt%02d" % i, (rng.standard_normal((int(t), F)) * 2).astype(np.float32)) for i, t in enumerate(rng.integers(90, 160, 9))]
    lens = [20, 35, 27, 20, 31]                                  # one chunk length per minibatch (create_egs.py:508-513)
    lines = []
    slots = [(mb, k) for mb in range(count) for k in range(B)]
    rng.shuffle(slots)
    for n, (mb, k) in enumerate(slots):
        u = int(rng.integers(0, len(utts)))
        off = int(rng.integers(0, utts[u][1].shape[0] - lens[mb]))
        lines.append("%s %d %d %d %d %d" % (utts[u][0], mb, 100 + mb, off, lens[mb], int(rng.integers(0, 7))))
    lines.sort(key=lambda l: l.split()[0])                       # ranges files are grouped by utterance
    tmp = tempfile.mkdtemp()
    ark, scp, rng_file = os.path.join(tmp, "feats.ark"), os.path.join(tmp, "feats.scp"), os.path.join(tmp, "ranges.1")
    used = set(l.split()[0] for l in lines)      # the recipe filters the scp to the archive's utterances (train_dnn.py:259)
    utts = [(k, m) for k, m in utts if k in used]
    with open(ark, "wb") as f, open(scp, "wt") as g:
        for k, m in utts:
            f.write((k + " ").encode())
            g.write("%s %s:%d\n" % (k, ark, f.tell()))
            ref_io.write_mat(f, m)
    open(rng_file, "wt").write("\n".join(lines) + "\n")
    res = {}
    for tag, shuffle, seed in (("plain", False, 0), ("shuffled", True, 11)):
        u2c, info = ex.process_range_file(rng_file, count, B)
        data, labels = ex.load_ranges_data(u2c, info, B, scp, F)
        if seed:
            np.random.seed(seed)
        if shuffle:
            perm = np.random.permutation(np.arange(count))
            data, labels = data[perm], labels[perm]
        dl = ex.DataLoader(data, labels, False)
        assert dl.count == count
        for i in range(count):
            d, l = dl.pop()
            res["%s_data_%d" % (tag, i)] = d
            res["%s_labels_%d" % (tag, i)] = l
        assert dl.pop() == (None, None)
    np.savez_compressed(out, ranges="\n".join(lines) + "\n", keys=np.array([k for k, _ in utts]),
                        **{"feat_%d" % i: m for i, (_, m) in enumerate(utts)}, count=count, minibatch_size=B, feat_dim=F, **res)
    print("egs_ranges: %d lines, %d minibatches x %d, served orders recorded (plain, shuffled seed 11)" % (len(lines), count, B))


def main():
    # XV_GOLDEN_OUT: write somewhere else (tests/test_golden_reproducible.py regenerates into a scratch directory and compares)
    out_dir = os.environ.get("XV_GOLDEN_OUT", HERE)
    ref_io, ref_models, Session = import_reference()
    if sys.argv[1:] == ["schedules"]:            # leaves the other (byte-stable) fixtures untouched
        golden_schedules(os.path.join(out_dir, "schedules.npz"))
        return
    if sys.argv[1:] == ["forward"]:
        golden_forward_refgraph(ref_io, ref_models, os.path.join(out_dir, "forward_refgraph.npz"))
        return
    if sys.argv[1:] == ["train"]:
        golden_train_refgraph(ref_models, os.path.join(out_dir, "train_refgraph.npz"))
        return
    if sys.argv[1:] == ["egs_ranges"]:
        golden_egs_ranges(os.path.join(out_dir, "egs_ranges.npz"))
        return
    golden_egs_ranges(os.path.join(out_dir, "egs_ranges.npz"))
    golden_schedules(os.path.join(out_dir, "schedules.npz"))
    golden_ark_io(ref_io, os.path.join(out_dir, "ark_io.npz"))
    golden_make_embedding(ref_io, ref_models, Session, os.path.join(out_dir, "make_embedding.npz"))
    golden_forward_refgraph(ref_io, ref_models, os.path.join(out_dir, "forward_refgraph.npz"))
    golden_train_refgraph(ref_models, os.path.join(out_dir, "train_refgraph.npz"))


if __name__ == "__main__":
    main()
