"""Read the reference's TensorFlow-1 checkpoints without TensorFlow.

``Model.save_model`` of the reference calls ``tf.train.Saver().save(sess, '<dir>/model')``
(local/tf/models.py:134-137), which writes a *tensor bundle*:

* ``model.index``                 -- an SSTable (LevelDB table format): sorted ``key -> value`` entries in
                                     prefix-compressed blocks; key "" holds a BundleHeaderProto, every other key
                                     is a variable name (``frame_level_info_layer-0/w``, no ``:0``) whose value is
                                     a BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c};
* ``model.data-00000-of-00001``   -- the raw little-endian tensor bytes, addressed by (offset, size);
* ``model.meta``                  -- the MetaGraphDef (not needed: the graph is re-stated by the kernels).

This module parses those two published formats directly (varints, block restarts, footer magic
0xdb4775248b80fb57; protobuf wire format for the two tiny messages; optional snappy blocks) and maps the
variables onto the TF-name-keyed weight dict the extractor loads.  What a checkpoint does NOT record is the
topology's dilation / activation (graph attributes): those come from the model CLASS NAME, which the
reference's driver stores next to the model (``<nnet_dir>/model_name.txt``, local/tf/train_dnn.py:495).

PARITY NOTE: TensorFlow cannot run here, so there is no reference-written checkpoint to pin against; the
reader is tested against a bundle writer that follows the same published layouts (tests/test_tf_checkpoint.py),
including multi-block tables, prefix-compressed keys, Adam slot variables and snappy-compressed blocks.
"""
import os
import struct

import numpy as np

from . import topology as tp

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 19: np.dtype("<f2")}


# ------------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------------
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _snappy_decompress(data):
    """Minimal snappy (raw format) decoder: varint length, then literal / copy elements."""
    n, pos = _varint(data, 0)
    out = bytearray()
    while pos < len(data):
        tag = data[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                  # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += data[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                  # copy, 1-byte offset
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | data[pos]
            pos += 1
        elif kind == 2:                                # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 2], "little")
            pos += 2
        else:                                          # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                            # overlapping copies are legal
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


def _read_block(buf, offset, size):
    """Block contents (without the 5-byte trailer), decompressed if needed."""
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    if ctype == 0:
        return raw
    if ctype == 1:
        return _snappy_decompress(raw)
    raise ValueError("unsupported table block compression type %d" % ctype)


def _block_entries(block):
    """Yield (key, value) from one table block (prefix-compressed keys, restart array at the end)."""
    (num_restarts,) = struct.unpack_from("<I", block, len(block) - 4)
    limit = len(block) - 4 - 4 * num_restarts
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path):
    """All (key, value) pairs of an SSTable file, in key order."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError("'%s' is not a TensorFlow/LevelDB table (bad magic)" % path)
    footer = buf[-48:]
    _, pos = _varint(footer, 0)            # metaindex handle: offset
    _, pos = _varint(footer, pos)          #                   size
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, idx_off, idx_size)):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        out.extend(_block_entries(_read_block(buf, off, size)))
    return out


def _parse_proto(buf):
    """Flat protobuf wire parse -> {field: [values]} (varint / 64-bit / length-delimited / 32-bit)."""
    out = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _entry(value):
    """BundleEntryProto -> (dtype enum, shape tuple, shard_id, offset, size)."""
    m = _parse_proto(value)
    dims = []
    for shape in m.get(2, []):
        for dim in _parse_proto(shape).get(2, []):
            size = _parse_proto(dim).get(1, [0])[0]
            dims.append(size if size < (1 << 63) else size - (1 << 64))
    g = lambda f: m.get(f, [0])[0]
    return g(1), tuple(dims), g(3), g(4), g(5)


# ------------------------------------------------------------------------------------------------
# bundle -> arrays -> weight dict
# ------------------------------------------------------------------------------------------------
def read_bundle(prefix):
    """{variable name: ndarray} for the tensor bundle ``<prefix>.index`` + ``<prefix>.data-*``."""
    entries = read_table(prefix + ".index")
    if not entries or entries[0][0] != b"":
        raise ValueError("'%s.index' has no bundle header" % prefix)
    header = _parse_proto(entries[0][1])
    num_shards = header.get(1, [1])[0]
    if header.get(2, [0])[0] != 0:
        raise ValueError("big-endian tensor bundles are not supported")
    shards = {}
    out = {}
    for key, value in entries[1:]:
        dtype, shape, shard, offset, size = _entry(value)
        if dtype not in _DTYPES:
            continue                                    # strings / resources: never model weights
        if shard not in shards:
            shards[shard] = np.memmap("%s.data-%05d-of-%05d" % (prefix, shard, num_shards), dtype=np.uint8, mode="r")
        dt = _DTYPES[dtype]
        count = int(np.prod(shape)) if shape else 1
        if count * dt.itemsize != size:
            raise ValueError("bundle entry '%s': size %d does not match shape %s" % (key.decode(), size, shape))
        out[key.decode()] = np.frombuffer(shards[shard][offset:offset + size].tobytes(), dtype=dt).reshape(shape)
    return out


def guess_class_name(model_dir):
    """The class name a deployment states: XVECTOR_MODEL_CLASS, else <nnet_dir>/model_name.txt, which the driver writes
    (train_dnn.py:495; model dirs are its children model_0, model_final, ...).  None when neither exists."""
    env = os.environ.get("XVECTOR_MODEL_CLASS")
    if env:
        return env
    for d in (model_dir, os.path.dirname(os.path.abspath(model_dir))):
        p = os.path.join(d, "model_name.txt")
        if os.path.exists(p):
            name = open(p).read().strip()
            if name:
                return name
    return None


def _is_slot(name):
    """Adam slots / power accumulators tf.train.AdamOptimizer adds to the checkpoint (models.py:112,134-137)."""
    return name.endswith("/Adam") or name.endswith("/Adam_1") or name in ("beta1_power", "beta2_power")


def extraction_signature(topo):
    """What of a topology changes the x-vector: everything else (dropout, L2 penalty, initialiser, head) is training-only."""
    return (tuple(topo["kernel_sizes"]), tuple(topo["layer_sizes"]), tuple(topo["embedding_sizes"]),
            topo.get("activation", "relu"), topo.get("pooling", "stats"))


def infer_signature(arrays, metagraph=b""):
    """The same tuple read off a checkpoint: kernel sizes / widths from the variable shapes, PReLU and attention from their
    variables; ReLU vs LeakyReLU only shows in the MetaGraphDef ``model.meta`` -- the bundle alone cannot tell them apart.
    ``tf.nn.leaky_relu(h, alpha=0.2, name='lrelu')`` (models.py:912) is a fused ``LeakyRelu`` op from TF 1.13 on and, in the
    TensorFlow of the reference's time (1.4-1.12), a COMPOSITE of ``mul`` + ``Maximum`` under the name scope ``lrelu`` -- no op
    type and no node name contains "LeakyRelu" there -- so both the op token and the scope token ``<layer>/lrelu`` count as
    LeakyReLU evidence; ``tf.nn.relu(h, name="relu")`` (models.py:64) leaves the node ``<layer>/relu``.  A graph that shows
    neither is inconclusive (None): an absence is not evidence."""
    kernels, widths, embeds = [], [], []
    i = 0
    while "frame_level_info_layer-%d/w" % i in arrays:
        k, _, cout = arrays["frame_level_info_layer-%d/w" % i].shape
        kernels.append(int(k)); widths.append(int(cout))
        i += 1
    j = 0
    while "embed_layer-%d/w" % j in arrays:
        embeds.append(int(arrays["embed_layer-%d/w" % j].shape[1]))
        j += 1
    if any(n.endswith("/prelu/prelu") for n in arrays):
        act = "prelu"
    elif b"LeakyRelu" in metagraph or b"frame_level_info_layer-0/lrelu" in metagraph:
        act = "lrelu"
    elif b"frame_level_info_layer-0/relu" in metagraph:
        act = "relu"
    else:
        act = None                   # no usable graph: ReLU and LeakyReLU checkpoints look the same
    pooling = "attention" if any(n.startswith("attention/") for n in arrays) else "stats"
    return (tuple(kernels), tuple(widths), tuple(embeds), act, pooling)


def class_for_signature(sig):
    """First class of the topology table with this extraction signature (classes that differ only in training-time
    settings share their x-vectors), or None."""
    for name, topo in tp.TOPOLOGIES.items():
        if extraction_signature(topo) == sig:
            return name
    return None


def weights_from_bundle(arrays, class_name):
    """Map bundle variables to (weights keyed by TF names with ':0', topology, num_classes, feat_dim).  The variable SET
    must be the one the class defines -- PReLU slopes or attention parameters the class would silently ignore are refused --
    and kernel sizes / widths are cross-checked against the variable shapes."""
    from . import weights as wio
    topo = tp.get(class_name)
    w = {}
    for name, arr in arrays.items():
        if _is_slot(name):
            continue
        w[name + ":0"] = np.ascontiguousarray(arr, dtype=np.float32)
    n_layers = len(topo["layer_sizes"])
    for i in range(n_layers):
        key = "frame_level_info_layer-%d/w:0" % i
        if key not in w:
            raise KeyError("checkpoint has no variable '%s'" % key[:-2])
        k, cin, cout = w[key].shape
        if k != topo["kernel_sizes"][i] or cout != topo["layer_sizes"][i]:
            raise ValueError("%s has shape %s but class %s expects kernel %d, %d channels (wrong model class? set "
                             "XVECTOR_MODEL_CLASS or model_name.txt)" % (key, w[key].shape, class_name,
                                                                           topo["kernel_sizes"][i], topo["layer_sizes"][i]))
    if topo["activation"] == "prelu" and "frame_level_info_layer-0/prelu/prelu:0" not in w:
        raise ValueError("class %s expects PReLU variables, the checkpoint has none" % class_name)
    expected = set(wio.expected_names(topo))
    scopes = ("frame_level_info_layer-", "embed_layer-", "attention/", "output/")
    stray = sorted(n for n in w if n.startswith(scopes) and n not in expected)
    if stray:
        raise ValueError("the checkpoint holds variables class %s does not define (%s): it was written by another model "
                         "class -- set XVECTOR_MODEL_CLASS or model_name.txt" % (class_name, ", ".join(stray[:4])))
    feat_dim = int(w["frame_level_info_layer-0/w:0"].shape[1])
    num_classes = int(w["output/w:0"].shape[1]) if "output/w:0" in w else 0
    return w, topo, num_classes, feat_dim


UNDERFLOWED_STEP_COUNT = 1000000


def optimizer_state_from_bundle(arrays):
    """Adam state of a reference-written checkpoint in the form weights.load_optimizer_state returns: ``<var>/Adam`` is the
    first moment, ``<var>/Adam_1`` the second.  TF keeps the POWERS ``beta1_power = 0.9**t`` / ``beta2_power = 0.999**t`` as
    float32 scalars, not the count: 0.9**t is denormal from t ~ 830 and zero from t ~ 985 -- any checkpoint past its first
    few hundred steps -- so the count is read off ``beta2_power`` (usable to t ~ 87 k), from ``beta1_power`` only while that is
    a normal float32, and once both have underflowed it is "large": the bias corrections 1 - beta**t are 1 by then, which is
    what TF itself computes from the zero powers (t = 0 would re-apply the first steps' correction of 0.15-0.3 x the learning
    rate for hundreds of steps).  None when the bundle has no slots (a model written before any training step)."""
    m = {n[:-len("/Adam")] + ":0": np.asarray(a, np.float32) for n, a in arrays.items() if n.endswith("/Adam")}
    v = {n[:-len("/Adam_1")] + ":0": np.asarray(a, np.float32) for n, a in arrays.items() if n.endswith("/Adam_1")}
    if not m or "beta1_power" not in arrays:
        return None
    tiny = float(np.finfo(np.float32).tiny)
    p1 = float(np.asarray(arrays["beta1_power"]).reshape(-1)[0])
    p2 = float(np.asarray(arrays["beta2_power"]).reshape(-1)[0]) if "beta2_power" in arrays else None
    if p1 >= 1.0 and (p2 is None or p2 >= 1.0):
        t = 0                                                   # slots exist but no step was taken
    elif p2 is not None and tiny <= p2 < 1.0:
        t = int(round(np.log(p2) / np.log(0.999)))
        if tiny <= p1 < 1.0 and t < 400:                        # few steps: 0.9**t resolves the count better than 0.999**t
            t = int(round(np.log(p1) / np.log(0.9)))
    elif tiny <= p1 < 1.0:
        t = int(round(np.log(p1) / np.log(0.9)))
    else:
        t = UNDERFLOWED_STEP_COUNT                              # both powers underflowed: the bias corrections are 1
    return dict(t=t, m=m, v=v)


def load_tf_model_dir(model_dir, class_name=None):
    """-> (weights, meta) like weights.load_model_dir, straight from a TF checkpoint directory.  The class comes from the
    caller / XVECTOR_MODEL_CLASS / model_name.txt and must agree with what the checkpoint itself shows (variables +
    MetaGraphDef); without a stated class the checkpoint's own evidence picks it, and that is logged."""
    import logging
    prefix = os.path.join(model_dir, "model")
    arrays = read_bundle(prefix)
    meta_path = os.path.join(model_dir, "model.meta")
    metagraph = open(meta_path, "rb").read() if os.path.exists(meta_path) else b""
    seen = infer_signature(arrays, metagraph)
    class_name = class_name or guess_class_name(model_dir)
    if class_name is None:
        if seen[3] is None:
            raise ValueError("'%s': no model class stated (XVECTOR_MODEL_CLASS / model_name.txt) and model.meta holds no graph "
                             "to tell a ReLU from a LeakyReLU checkpoint -- refusing to guess" % model_dir)
        class_name = class_for_signature(seen)
        if class_name is None:
            raise ValueError("'%s': no model class stated (XVECTOR_MODEL_CLASS / model_name.txt) and the checkpoint matches "
                             "none of the known classes: kernels %s, widths %s, embeddings %s, %s, %s pooling" % ((model_dir,) + seen))
        logging.getLogger(__name__).warning("no model class stated for '%s' (XVECTOR_MODEL_CLASS / model_name.txt): using %s, "
                                            "inferred from the checkpoint's variables and graph", model_dir, class_name)
    else:
        stated = extraction_signature(tp.get(class_name))[3]
        # the variable set is checked in weights_from_bundle; the one thing only the graph shows is LeakyReLU vs ReLU
        if seen[3] in ("relu", "lrelu") and stated in ("relu", "lrelu") and stated != seen[3]:
            raise ValueError("'%s': class %s uses %s but the checkpoint's graph (model.meta) shows %s nodes in the frame-level "
                             "layers (LeakyRelu / lrelu vs relu) -- wrong model class"
                             % (model_dir, class_name, stated, "LeakyReLU" if seen[3] == "lrelu" else "ReLU"))
        elif seen[3] is None and stated in ("relu", "lrelu"):
            logging.getLogger(__name__).warning("'%s': model.meta shows neither relu nor lrelu nodes; taking the stated class %s "
                                                "(%s) at its word", model_dir, class_name, stated)
    w, topo, num_classes, feat_dim = weights_from_bundle(arrays, class_name)
    meta = dict(format="tensorflow-checkpoint", model_class=class_name, topology=topo, num_classes=num_classes,
                feat_dim=feat_dim)
    return w, meta
