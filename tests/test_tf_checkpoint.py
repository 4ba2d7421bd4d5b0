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
