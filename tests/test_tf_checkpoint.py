"""TF1 checkpoint reader against a bundle WRITER that follows the published layouts: LevelDB table
format (prefix-compressed entries, restart arrays, block trailers, index block, 48-byte footer with magic) and
the BundleHeaderProto / BundleEntryProto wire encodings.  TensorFlow cannot run here, so no reference-written file
exists to pin against (stated in tf_checkpoint.py); this pins the reader against an independent statement of the format.
"""
import os
import struct

import numpy as np
import pytest

from xvector_amd import synthetic, tf_checkpoint, topology, weights as wio


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    return _varint((num << 3) | wt) + payload


def _entry_proto(dtype, shape, offset, size):
    dims = b"".join(_field(2, 2, _varint(len(d)) + d) for d in (_field(1, 0, _varint(s)) for s in shape))
    return (_field(1, 0, _varint(dtype)) + _field(2, 2, _varint(len(dims)) + dims) + _field(4, 0, _varint(offset)) +
            _field(5, 0, _varint(size)) + _field(6, 5, struct.pack("<I", 0xDEADBEEF)))


def _snappy_literals(data):
    """A valid snappy stream made of literal elements only."""
    out = bytearray(_varint(len(data)))
    pos = 0
    while pos < len(data):
        chunk = data[pos:pos + 60]
        out.append((len(chunk) - 1) << 2)
        out += chunk
        pos += len(chunk)
    return bytes(out)


def _build_block(entries, restart_interval=4):
    buf = bytearray()
    restarts = []
    prev = b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        buf += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_bundle(prefix, arrays, block_entries=5, snappy=False):
    """arrays: {name: ndarray}.  Writes <prefix>.index and <prefix>.data-00000-of-00001."""
    data = bytearray()
    items = [(b"", _field(1, 0, _varint(1)) + _field(3, 2, _varint(2) + _field(1, 0, _varint(1))))]      # header: 1 shard, version
    for name in sorted(arrays):
        a = np.asarray(arrays[name], order="C")
        dtype = {np.dtype("float32"): 1, np.dtype("int32"): 3, np.dtype("float64"): 2}[a.dtype]
        items.append((name.encode(), _entry_proto(dtype, a.shape, len(data), a.nbytes)))
        data += a.tobytes()
    items.sort(key=lambda kv: kv[0])
    out = bytearray()
    index_entries = []

    def emit(block):
        payload = _snappy_literals(block) if snappy else block
        off = len(out)
        out.extend(payload + bytes([1 if snappy else 0]) + b"\0\0\0\0")                 # trailer: type + crc (unchecked)
        return _varint(off) + _varint(len(payload))

    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        index_entries.append((chunk[-1][0] + b"\xff", emit(_build_block(chunk))))          # separator >= last key
    meta_handle = emit(_build_block([]))
    index_handle = emit(_build_block(index_entries, restart_interval=1))
    footer = meta_handle + index_handle
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", tf_checkpoint.TABLE_MAGIC)
    out += footer
    with open(prefix + ".index", "wb") as f:
        f.write(out)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)


def _tf_style_arrays(w, with_adam=True):
    arrays = {k[:-2]: v for k, v in w.items()}                     # TF bundle keys carry no ':0'
    if with_adam:
        arrays["beta1_power"] = np.array(0.9, np.float32).reshape(())
        arrays["beta2_power"] = np.array(0.999, np.float32).reshape(())
        for k in list(arrays):
            if k.endswith("/w") or k.endswith("/b"):
                arrays[k + "/Adam"] = np.zeros_like(arrays[k])
                arrays[k + "/Adam_1"] = np.ones_like(arrays[k])
    return arrays


@pytest.mark.parametrize("snappy", [False, True])
def test_bundle_roundtrip(tmp_path, snappy):
    topo = synthetic.SMALL_TOPOLOGY
    w = synthetic.trained_like(topo, 5, num_classes=8, seed=3)
    arrays = _tf_style_arrays(w)
    arrays["global_step"] = np.array([7], np.int32)
    write_bundle(str(tmp_path / "model"), arrays, block_entries=5, snappy=snappy)
    back = tf_checkpoint.read_bundle(str(tmp_path / "model"))
    assert sorted(back) == sorted(arrays)
    for k in arrays:
        assert back[k].dtype == arrays[k].dtype and back[k].shape == arrays[k].shape and np.array_equal(back[k], arrays[k])


def test_load_model_dir_accepts_a_tf_checkpoint_directory(tmp_path):
    """The reference's directory layout: <nnet>/model_name.txt + <nnet>/model_final/{model.meta(protobuf),model.index,
    model.data-*,done} (train_dnn.py:495, models.py:130-141) loads without TensorFlow, dilations from the class name."""
    topo = topology.get("ModelWithoutDropoutTdnn")
    topo["layer_sizes"] = [32, 32, 32, 32, 48]; topo["embedding_sizes"] = [16, 16]
    w = synthetic.trained_like(topo, 23, num_classes=10, seed=4)
    nnet = tmp_path / "xvector_nnet"
    mdir = nnet / "model_final"
    mdir.mkdir(parents=True)
    (nnet / "model_name.txt").write_text("ModelWithoutDropoutTdnn\n")
    (mdir / "model.meta").write_bytes(b"\x0a\x8f\x01\x0a\x0bplaceholder-metagraph")       # not JSON: a protobuf blob
    (mdir / "done").write_text("done")
    write_bundle(str(mdir / "model"), _tf_style_arrays(w))
    assert wio.is_correct_model_dir(str(mdir))
    # widths differ from the class defaults -> the shape cross-check must complain ...
    with pytest.raises(ValueError):
        wio.load_model_dir(str(mdir))
    # ... and pass for the real default-width topology
    topo = topology.get("ModelWithoutDropoutTdnn")
    w = synthetic.reference_init(topo, 23, 10, seed=4)
    write_bundle(str(mdir / "model"), _tf_style_arrays(w), block_entries=7)
    got, meta = wio.load_model_dir(str(mdir))
    assert meta["model_class"] == "ModelWithoutDropoutTdnn" and meta["topology"]["dilations"] == [1, 2, 3, 1, 1]
    assert meta["feat_dim"] == 23 and meta["num_classes"] == 10
    assert sorted(got) == sorted(w) and all(np.array_equal(got[k], w[k]) for k in w)


def test_bad_files_are_rejected(tmp_path):
    p = tmp_path / "model.index"
    p.write_bytes(b"\0" * 100)
    with pytest.raises(ValueError):
        tf_checkpoint.read_table(str(p))
    with pytest.raises(ValueError):
        tf_checkpoint._snappy_decompress(bytes([5, 0x05, 1]))          # copy before any literal
    (tmp_path / "model.meta").write_bytes(b"\x0a\x01x")
    os.remove(str(p))
    with pytest.raises(IOError):
        wio.load_model_dir(str(tmp_path))


def _tf_dir(tmp_path, class_written, meta_bytes, name_file=None, seed=5):
    """A directory as the reference's Saver leaves it for a model of class ``class_written`` (default widths)."""
    topo = topology.get(class_written)
    w = synthetic.reference_init(topo, 23, 10, seed=seed)
    nnet = tmp_path / ("nnet_%s_%d" % (class_written, len(list(tmp_path.iterdir()))))
    mdir = nnet / "model_7"
    mdir.mkdir(parents=True)
    if name_file:
        (nnet / "model_name.txt").write_text(name_file + "\n")
    (mdir / "model.meta").write_bytes(meta_bytes)
    (mdir / "done").write_text("done")
    write_bundle(str(mdir / "model"), _tf_style_arrays(w))
    return str(mdir), w


GRAPH_BODY = b"\x0a\x1fframe_level_info_layer-0/conv1d\x12\x06Conv2D"
GRAPH_RELU = GRAPH_BODY + b"\x0a\x1dframe_level_info_layer-0/relu\x12\x04Relu"                 # tf.nn.relu(h, name="relu")
GRAPH_LRELU = GRAPH_BODY + b"\x0a\x1eframe_level_info_layer-0/lrelu\x12\x09LeakyRelu"         # TF >= 1.13: fused op
# TF 1.4-1.12 (the reference's time): leaky_relu is mul + Maximum under the scope 'lrelu' -- no "LeakyRelu" anywhere
GRAPH_LRELU_COMPOSITE = GRAPH_BODY + b"\x0a\x22frame_level_info_layer-0/lrelu/mul\x12\x03Mul" \
    b"\x0a\x26frame_level_info_layer-0/lrelu/Maximum\x12\x07Maximum"


def test_wrong_or_missing_model_class_is_refused(tmp_path, monkeypatch, caplog):
    """A PReLU / LeakyReLU / attention checkpoint has the conv shapes of the default class: loading it as ReLU would give
    wrong x-vectors silently.  Stray variables are refused, the graph decides ReLU vs LeakyReLU, and without any stated
    class and without a graph nothing is guessed."""
    monkeypatch.delenv("XVECTOR_MODEL_CLASS", raising=False)
    # PReLU checkpoint, directory claims the default class -> stray prelu variables
    d, _ = _tf_dir(tmp_path, "ModelWithoutDropoutPRelu", GRAPH_RELU, name_file="ModelWithoutDropout")
    with pytest.raises(ValueError, match="does not define"):
        wio.load_model_dir(d)
    # same checkpoint, class stated by the environment -> fine
    monkeypatch.setenv("XVECTOR_MODEL_CLASS", "ModelWithoutDropoutPRelu")
    got, meta = wio.load_model_dir(d)
    assert meta["topology"]["activation"] == "prelu" and "frame_level_info_layer-3/prelu/prelu:0" in got
    monkeypatch.delenv("XVECTOR_MODEL_CLASS")
    # LeakyReLU checkpoint claimed to be ReLU: only the graph shows it
    d, _ = _tf_dir(tmp_path, "ModelL2LossWithoutDropoutLRelu", GRAPH_LRELU, name_file="ModelWithoutDropout")
    with pytest.raises(ValueError, match="LeakyRelu"):
        wio.load_model_dir(d)
    d, _ = _tf_dir(tmp_path, "ModelWithoutDropout", GRAPH_RELU, name_file="ModelL2LossWithoutDropoutLRelu")
    with pytest.raises(ValueError, match="LeakyRelu"):
        wio.load_model_dir(d)
    # the composite form of the reference's own TensorFlow: a correctly stated LeakyReLU class loads, a ReLU claim is refused,
    # and without a stated class it is inferred as LeakyReLU (not silently as ReLU)
    d, _ = _tf_dir(tmp_path, "ModelL2LossWithoutDropoutLRelu", GRAPH_LRELU_COMPOSITE, name_file="ModelL2LossWithoutDropoutLRelu")
    assert wio.load_model_dir(d)[1]["topology"]["activation"] == "lrelu"
    d, _ = _tf_dir(tmp_path, "ModelL2LossWithoutDropoutLRelu", GRAPH_LRELU_COMPOSITE, name_file="ModelWithoutDropout")
    with pytest.raises(ValueError, match="LeakyRelu"):
        wio.load_model_dir(d)
    d, _ = _tf_dir(tmp_path, "ModelL2LossWithoutDropoutLRelu", GRAPH_LRELU_COMPOSITE)
    assert wio.load_model_dir(d)[1]["topology"]["activation"] == "lrelu"
    # a graph that names the layers but shows neither node is inconclusive: a stated class is taken at its word (warning) ...
    d, _ = _tf_dir(tmp_path, "ModelL2LossWithoutDropoutLRelu", GRAPH_BODY, name_file="ModelL2LossWithoutDropoutLRelu")
    with caplog.at_level("WARNING"):
        assert wio.load_model_dir(d)[1]["topology"]["activation"] == "lrelu"
    assert "at its word" in caplog.text
    # no class stated: the checkpoint's own evidence picks it (and says so) ...
    d, _ = _tf_dir(tmp_path, "ModelL2LossWithoutDropoutLRelu", GRAPH_LRELU)
    with caplog.at_level("WARNING"):
        _, meta = wio.load_model_dir(d)
    assert meta["topology"]["activation"] == "lrelu" and "inferred from the checkpoint" in caplog.text
    d, _ = _tf_dir(tmp_path, "ModelWithoutDropoutTdnn", GRAPH_RELU)
    assert wio.load_model_dir(d)[1]["topology"]["dilations"] == [1, 2, 3, 1, 1]
    d, _ = _tf_dir(tmp_path, "ModelL2LossWithoutDropoutLReluAttention", GRAPH_LRELU)
    assert wio.load_model_dir(d)[1]["topology"]["pooling"] == "attention"
    # ... unless there is no graph to tell ReLU from LeakyReLU
    d, _ = _tf_dir(tmp_path, "ModelWithoutDropout", b"\x0a\x03abc")
    with pytest.raises(ValueError, match="refusing to guess"):
        wio.load_model_dir(d)


def test_adam_slots_of_a_tf_checkpoint_resume(tmp_path):
    """<var>/Adam, <var>/Adam_1 and beta1_power of a reference-written checkpoint become the optimizer state the trainer
    resumes from (the reference's Saver.restore does the same, models.py:232-236)."""
    topo = topology.get("ModelWithoutDropout")
    w = synthetic.reference_init(topo, 23, 10, seed=2)
    arrays = {k[:-2]: v for k, v in w.items()}
    rng = np.random.default_rng(0)
    for k in list(arrays):
        if k.rsplit("/", 1)[-1] in ("w", "b", "gamma", "beta"):
            arrays[k + "/Adam"] = rng.standard_normal(arrays[k].shape).astype(np.float32)
            arrays[k + "/Adam_1"] = rng.random(arrays[k].shape).astype(np.float32)
    arrays["beta1_power"] = np.array(0.9 ** 37, np.float32)
    arrays["beta2_power"] = np.array(0.999 ** 37, np.float32)
    mdir = tmp_path / "model_3"
    mdir.mkdir()
    (mdir / "model.meta").write_bytes(GRAPH_RELU)
    (mdir / "done").write_text("done")
    (mdir / "model_name.txt").write_text("ModelWithoutDropout")
    write_bundle(str(mdir / "model"), arrays)
    adam = wio.load_optimizer_state(str(mdir))
    assert adam["t"] == 37
    # long runs: 0.9**t underflows float32 (denormal from ~830, zero from ~985) -- the count comes from beta2_power, and when
    # both powers are gone it is "large" (bias corrections 1), never 0
    from xvector_amd import tf_checkpoint
    for t_true in (5, 399, 900, 5000, 60000):
        got = tf_checkpoint.optimizer_state_from_bundle(dict(arrays, beta1_power=np.float32(0.9) ** np.float32(t_true),
                                                             beta2_power=np.float32(0.999 ** t_true)))["t"]
        assert abs(got - t_true) <= max(1, t_true // 200), (t_true, got)
    gone = tf_checkpoint.optimizer_state_from_bundle(dict(arrays, beta1_power=np.float32(0), beta2_power=np.float32(0)))
    assert gone["t"] == tf_checkpoint.UNDERFLOWED_STEP_COUNT
    assert tf_checkpoint.optimizer_state_from_bundle(dict(arrays, beta1_power=np.float32(1), beta2_power=np.float32(1)))["t"] == 0
    assert np.array_equal(adam["m"]["embed_layer-0/w:0"], arrays["embed_layer-0/w/Adam"])
    assert np.array_equal(adam["v"]["frame_level_info_layer-2/gamma:0"], arrays["frame_level_info_layer-2/gamma/Adam_1"])
    assert set(adam["m"]) == set(adam["v"]) == {k + ":0" for k in arrays if k + "/Adam" in arrays}
    # a model written before the first step has no slots: fresh optimizer
    write_bundle(str(mdir / "model"), {k[:-2]: v for k, v in w.items()})
    assert wio.load_optimizer_state(str(mdir)) is None
