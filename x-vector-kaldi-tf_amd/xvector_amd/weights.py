"""Model-directory weight container.

The reference persists a TF1 checkpoint (``model.meta/.index/.data-*`` + a ``done`` marker,
``local/tf/models.py:130-141``) and every caller only tests that ``model.meta`` and ``done`` exist and
are non-empty (``local/tf/ze_utils.py:561-567``, ``local/tf/extract_embedding.py:88``,
``local/tf/extract_xvectors.sh:44``).  This build keeps that file contract and stores

* ``model.meta``          -- JSON: format tag, class name, topology, num_classes, feat_dim
* ``model.weights.npz``   -- float32 arrays keyed by the TF variable names the reference's h5 export
                             enumerates (``local/tf/models.py:199-213``):
                             ``frame_level_info_layer-{i}/{w,b,gamma,beta,mean,variance}:0``,
                             ``embed_layer-{j}/{w,b,gamma,beta,mean,variance}:0``, ``output/{w,b}:0``,
                             optional ``.../prelu/prelu:0``, ``attention/{w,b,v}:0``
* ``done``                -- the marker the drivers look for.

A directory written by the reference itself (TF1 ``Saver``: ``model.meta`` protobuf + ``model.index`` +
``model.data-00000-of-00001``) is also accepted by ``load_model_dir``: see ``tf_checkpoint.py``.
"""
import json
import os

import numpy as np

FORMAT_TAG = "xvector-amd-weights-v1"
META = "model.meta"
WEIGHTS = "model.weights.npz"
DONE = "done"


def expected_names(topo):
    names = []
    for i in range(len(topo["layer_sizes"])):
        sc = "frame_level_info_layer-%d" % i
        names += ["%s/%s:0" % (sc, n) for n in ("w", "b", "gamma", "beta", "mean", "variance")]
        if topo.get("activation") == "prelu":
            names.append("%s/prelu/prelu:0" % sc)
    for j in range(len(topo["embedding_sizes"])):
        sc = "embed_layer-%d" % j
        names += ["%s/%s:0" % (sc, n) for n in ("w", "b", "gamma", "beta", "mean", "variance")]
        if topo.get("activation") == "prelu":
            names.append("%s/prelu/prelu:0" % sc)
    if topo.get("pooling", "stats") == "attention":
        names += ["attention/w:0", "attention/b:0", "attention/v:0"]          # models.py:1040-1043
    names += ["output/w:0", "output/b:0"]
    return names


def save_model_dir(output_dir, weights, topo, class_name, num_classes, feat_dim):
    os.makedirs(output_dir, exist_ok=True)
    missing = [n for n in expected_names(topo) if n not in weights]
    if missing:
        raise KeyError("weights missing for: %s" % ", ".join(missing))
    arrays = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
    tmp = os.path.join(output_dir, WEIGHTS + ".tmp.npz")
    np.savez(tmp, **arrays)
    os.replace(tmp, os.path.join(output_dir, WEIGHTS))
    meta = dict(format=FORMAT_TAG, model_class=class_name, topology=topo,
                num_classes=int(num_classes), feat_dim=int(feat_dim))
    with open(os.path.join(output_dir, META), "wt") as fid:
        json.dump(meta, fid, indent=1, sort_keys=True)
        fid.write("\n")
    with open(os.path.join(output_dir, DONE), "wt") as fid:      # models.py:138-139
        fid.write("done")


OPTIMIZER = "model.optimizer.npz"


def save_optimizer_state(output_dir, adam):
    """Adam slots (the reference's Saver stores them as <var>/Adam, <var>/Adam_1, beta{1,2}_power in the same
    checkpoint, models.py:134-137, so that the next iteration resumes them): {"t", "m": {name: arr}, "v": {...}}."""
    arrays = {"t": np.array(int(adam["t"]), np.int64)}
    for k, v in adam["m"].items():
        arrays["m/" + k] = np.asarray(v, np.float32)
    for k, v in adam["v"].items():
        arrays["v/" + k] = np.asarray(v, np.float32)
    tmp = os.path.join(output_dir, OPTIMIZER + ".tmp.npz")
    np.savez(tmp, **arrays)
    os.replace(tmp, os.path.join(output_dir, OPTIMIZER))


def load_optimizer_state(input_dir):
    """-> adam dict or None (fresh optimizer) when the directory has no optimizer state."""
    p = os.path.join(input_dir, OPTIMIZER)
    if not os.path.exists(p):
        # a checkpoint the reference's Saver wrote carries the slots inside the bundle (models.py:134-137)
        if os.path.exists(os.path.join(input_dir, "model.index")):
            from . import tf_checkpoint
            return tf_checkpoint.optimizer_state_from_bundle(tf_checkpoint.read_bundle(os.path.join(input_dir, "model")))
        return None
    adam = {"t": 0, "m": {}, "v": {}}
    with np.load(p) as z:
        for k in z.files:
            if k == "t":
                adam["t"] = int(z[k])
            else:
                adam[k[0]][k[2:]] = np.asarray(z[k], np.float32)
    return adam


TRAIN_VERDICT = "train_arithmetic.json"
TRAIN_VERDICT_REPROBE_EVERY = 10


def save_train_verdict(model_dir, verdict):
    """The arithmetic the training of this model runs in (trainer.select_trainer's verdict + how many iterations ago it was probed)."""
    with open(os.path.join(model_dir, TRAIN_VERDICT), "wt") as fid:
        json.dump(verdict, fid)


def load_train_verdict(model_dir):
    """-> verdict dict or None (no verdict, unreadable, or due for a fresh probe)."""
    try:
        with open(os.path.join(model_dir, TRAIN_VERDICT), "rt") as fid:
            v = json.load(fid)
        if v.get("selected") not in ("fp32", "bf16x3"):
            return None
        if v["selected"] == "bf16x3" and int(v.get("iterations_since_probe", 0)) >= TRAIN_VERDICT_REPROBE_EVERY:
            return None                                  # the weights have moved: admit bf16x3 again on real gradients
        return v
    except Exception:
        return None


def is_correct_model_dir(model_dir):
    """Same predicate as the reference's ``ze_utils.is_correct_model_dir`` (ze_utils.py:561-567)."""
    for name in (META, DONE):
        p = os.path.join(model_dir, name)
        if not os.path.exists(p) or os.path.getsize(p) == 0:
            return False
    return True


def _read_stored_npz(path):
    """{name: float32 array} of an UNCOMPRESSED ``.npz`` (what ``np.savez`` writes) as read-only views of the memory-mapped
    file: ``np.load`` runs every member through zipfile's CRC check and a copy (30 ms for this model's 25 MB -- a tenth of a
    50 k-utterance extraction), here the member payloads are located through the zip directory and used where they are.
    None when the file is not a plain stored archive of C-order arrays (the caller falls back to ``np.load``)."""
    import ast
    import mmap
    import struct
    import zipfile
    try:
        out = {}
        with open(path, "rb") as f:
            infos = zipfile.ZipFile(f).infolist()
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        for info in infos:
            if info.compress_type != zipfile.ZIP_STORED or not info.filename.endswith(".npy"):
                return None
            h = info.header_offset
            if mm[h:h + 4] != b"PK\x03\x04":
                return None
            nlen, elen = struct.unpack("<HH", mm[h + 26:h + 30])
            start = h + 30 + nlen + elen
            if mm[start:start + 6] != b"\x93NUMPY":
                return None
            if mm[start + 6] == 1:
                hlen, off = struct.unpack("<H", mm[start + 8:start + 10])[0], 10
            else:
                hlen, off = struct.unpack("<I", mm[start + 8:start + 12])[0], 12
            header = ast.literal_eval(mm[start + off:start + off + hlen].decode("latin1"))
            if header.get("fortran_order"):
                return None
            dtype, shape = np.dtype(header["descr"]), tuple(header["shape"])
            if dtype.hasobject:
                return None
            count = int(np.prod(shape)) if shape else 1
            arr = np.frombuffer(mm, dtype=dtype, count=count, offset=start + off + hlen).reshape(shape)
            out[info.filename[:-4]] = arr if arr.dtype == np.float32 else arr.astype(np.float32)
        return out
    except Exception:
        return None


def load_model_dir(input_dir):
    """-> (weights dict, meta dict).  Raises with a clear message on a TF checkpoint directory."""
    meta_path = os.path.join(input_dir, META)
    if not os.path.exists(meta_path):
        raise IOError("no %s in '%s'" % (META, input_dir))
    with open(meta_path, "rb") as fid:
        raw = fid.read()
    try:
        meta = json.loads(raw.decode("utf-8"))
        assert meta.get("format") == FORMAT_TAG
    except Exception:
        # not ours: a directory written by the reference's tf.train.Saver (model.meta = MetaGraphDef,
        # model.index + model.data-* = tensor bundle) is read directly, without TensorFlow
        if os.path.exists(os.path.join(input_dir, "model.index")):
            from . import tf_checkpoint
            weights, meta = tf_checkpoint.load_tf_model_dir(input_dir)
            missing = [n for n in expected_names(meta["topology"]) if n not in weights]
            if missing:
                raise KeyError("TF checkpoint in '%s' lacks variables: %s" % (input_dir, ", ".join(missing)))
            return weights, meta
        raise IOError("'%s' is neither an %s model directory nor a TensorFlow checkpoint directory (no model.index)"
                      % (input_dir, FORMAT_TAG))
    weights = _read_stored_npz(os.path.join(input_dir, WEIGHTS))
    if weights is None:
        with np.load(os.path.join(input_dir, WEIGHTS)) as z:
            weights = {k: np.asarray(z[k], dtype=np.float32) for k in z.files}
    missing = [n for n in expected_names(meta["topology"]) if n not in weights]
    if missing:
        raise KeyError("model dir '%s' lacks variables: %s" % (input_dir, ", ".join(missing)))
    return weights, meta
