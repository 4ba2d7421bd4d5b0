"""Kaldi ark/scp I/O twin vs golden bytes produced by the REFERENCE's own kaldi_io
(tests/golden/ark_io.npz, written by tests/golden/make_golden.py in the build container)."""
import gzip
import io
import os

import numpy as np
import pytest

import kaldi_io


@pytest.fixture(scope="module")
def g(golden):
    return golden("ark_io.npz")


def test_write_mat_bytes_identical_to_reference(g):
    bio = io.BytesIO()
    kaldi_io.write_mat(bio, g["fm"], key="utt-a.1")
    kaldi_io.write_mat(bio, g["dm"], key="utt_b/2")
    kaldi_io.write_mat(bio, np.zeros((0, 23), np.float32), key="empty")
    assert bio.getvalue() == g["mat_ark"].tobytes()


def test_write_vec_flt_bytes_identical_to_reference(g):
    bio = io.BytesIO()
    kaldi_io.write_vec_flt(bio, g["fv"], key="spk1")
    kaldi_io.write_vec_flt(bio, g["dv"], key="spk2")
    assert bio.getvalue() == g["vec_ark"].tobytes()
    # framing documented in SURVEY §8a-10: key, space, \0B, 'FV ', \4, uint32 dim, payload
    assert bio.getvalue()[:15] == b"spk1 \x00BFV \x04" + (512).to_bytes(4, "little")


def test_read_mat_ark_binary(g):
    got = list(kaldi_io.read_mat_ark(io.BytesIO(g["mat_ark"].tobytes())))
    assert [k for k, _ in got] == ["utt-a.1", "utt_b/2", "empty"]
    assert got[0][1].dtype == np.float32 and np.array_equal(got[0][1], g["fm"])
    assert got[1][1].dtype == np.float64 and np.array_equal(got[1][1], g["dm"])
    assert got[2][1].shape == (0, 23)


def test_read_vec_flt_ark(g):
    got = list(kaldi_io.read_vec_flt_ark(io.BytesIO(g["vec_ark"].tobytes())))
    assert [k for k, _ in got] == ["spk1", "spk2"]
    assert np.array_equal(got[0][1], g["fv"]) and got[0][1].dtype == np.float32
    assert np.array_equal(got[1][1], g["dv"]) and got[1][1].dtype == np.float64
    txt = list(kaldi_io.read_vec_flt_ark(io.BytesIO(g["vec_txt"].tobytes())))
    assert txt[0][0] == "vtx" and np.array_equal(txt[0][1], g["vec_txt_dec"])


def test_compressed_matrix_decodes_like_reference(g):
    (key, mat), = list(kaldi_io.read_mat_ark(io.BytesIO(g["cm_ark"].tobytes())))
    assert key == "cm1"
    assert mat.dtype == np.float32 and mat.shape == g["cm_dec"].shape
    assert np.array_equal(mat, g["cm_dec"])          # bit-exact with the reference decode


def test_compressed_cm2_cm3():
    """The two-byte / one-byte forms Kaldi's automatic method picks for matrices of at most 8 rows.  Tokens carry Kaldi's trailing space
    ("CM2 ", compressed-matrix.cc CompressedMatrix::Write -> WriteToken); the reference reader asserts 'CM ' and knows neither."""
    rows, cols = 3, 5
    hdr = np.array([-1.0, 2.0], "<f4").tobytes() + np.array([rows, cols], "<i4").tobytes()
    u16 = np.arange(rows * cols, dtype="<u2").reshape(rows, cols) * 4000
    m2 = kaldi_io.read_mat(io.BytesIO(b"\x00BCM2 " + hdr + u16.tobytes()))
    assert np.allclose(m2, -1.0 + 2.0 * u16 / 65535.0, atol=1e-6) and m2.shape == (rows, cols)
    u8 = (np.arange(rows * cols, dtype=np.uint8).reshape(rows, cols) * 17)
    m3 = kaldi_io.read_mat(io.BytesIO(b"\x00BCM3 " + hdr + u8.tobytes()))
    assert np.allclose(m3, -1.0 + 2.0 * u8 / 255.0, atol=1e-6)


def test_ascii_matrix(g):
    (key, mat), = list(kaldi_io.read_mat_ark(io.BytesIO(g["txt_ark"].tobytes())))
    assert key == "txt1" and mat.dtype == np.float32
    assert np.array_equal(mat, g["txt_dec"])


def test_errors_match_reference_types():
    with pytest.raises(kaldi_io.UnknownMatrixHeader):
        kaldi_io.read_mat(io.BytesIO(b"\x00BXM \x04\x01\x00\x00\x00\x04\x01\x00\x00\x00"))
    with pytest.raises(kaldi_io.UnknownVectorHeader):
        kaldi_io.read_vec_flt(io.BytesIO(b"\x00BXV \x04\x01\x00\x00\x00"))
    with pytest.raises(kaldi_io.UnsupportedDataType):
        kaldi_io.write_vec_flt(io.BytesIO(), np.zeros(3, np.int32), key="a")
    with pytest.raises(kaldi_io.UnsupportedDataType):
        kaldi_io.write_mat(io.BytesIO(), np.zeros((3, 2), np.float16), key="a")
    with pytest.raises(kaldi_io.BadInputFormat):
        kaldi_io.read_mat(io.BytesIO(b" [\n 1 2\n"))
    with pytest.raises(AssertionError):
        kaldi_io.read_key(io.BytesIO(b"bad!key "))
    assert kaldi_io.read_key(io.BytesIO(b"")) is None


def test_open_or_fd_specifiers_offsets_gz_and_pipes(tmp_path, g):
    ark = tmp_path / "feats.ark"
    ark.write_bytes(g["mat_ark"].tobytes())
    # ark: prefix and plain path
    for spec in ("ark:%s" % ark, str(ark), "ark,t:%s" % ark):
        with kaldi_io.open_or_fd(spec) as fd:
            assert kaldi_io.read_key(fd) == "utt-a.1"
    # path:offset (scp style): offset just past "utt-a.1 "
    m = kaldi_io.read_mat("%s:%d" % (ark, len("utt-a.1 ")))
    assert np.array_equal(m, g["fm"])
    # scp
    scp = tmp_path / "feats.scp"
    off2 = len("utt-a.1 ") + 2 + 3 + 10 + g["fm"].nbytes + len("utt_b/2 ")
    scp.write_text("utt-a.1 %s:%d\nutt_b/2 %s:%d\n" % (ark, len("utt-a.1 "), ark, off2))
    got = dict(kaldi_io.read_mat_scp(str(scp)))
    assert np.array_equal(got["utt-a.1"], g["fm"]) and np.array_equal(got["utt_b/2"], g["dm"])
    # gz
    gz = tmp_path / "feats.ark.gz"
    with gzip.open(str(gz), "wb") as f:
        f.write(g["mat_ark"].tobytes())
    assert [k for k, _ in kaldi_io.read_mat_ark(str(gz))] == ["utt-a.1", "utt_b/2", "empty"]
    # read pipe / write pipe
    assert [k for k, _ in kaldi_io.read_mat_ark("ark:cat %s |" % ark)] == ["utt-a.1", "utt_b/2", "empty"]
    out = tmp_path / "piped.ark"
    fd = kaldi_io.open_or_fd("ark:| cat > %s" % out, "wb")
    kaldi_io.write_vec_flt(fd, g["fv"], key="spk1")
    fd.close()
    for _ in range(100):
        if out.exists() and out.stat().st_size == 5 + 2 + 3 + 5 + g["fv"].nbytes:
            break
        import time
        time.sleep(0.05)
    assert np.array_equal(dict(kaldi_io.read_vec_flt_ark(str(out)))["spk1"], g["fv"])
    # an already opened stream passes through untouched
    bio = io.BytesIO(b"x")
    assert kaldi_io.open_or_fd(bio) is bio


def test_table_writer_ark_scp(tmp_path, g):
    ark, scp = str(tmp_path / "x.ark"), str(tmp_path / "x.scp")
    with kaldi_io.TableWriter(ark, scp) as w:
        kaldi_io.write_vec_flt(w, g["fv"], key="a")
        kaldi_io.write_vec_flt(w, g["fv"] * 2, key="b")
    lines = open(scp).read().splitlines()
    assert lines[0] == "a %s:2" % ark
    got = dict(kaldi_io.read_vec_flt_scp(scp))
    assert np.array_equal(got["a"], g["fv"]) and np.array_equal(got["b"], g["fv"] * 2)
    assert os.path.getsize(ark) == 2 * (2 + 2 + 3 + 5 + g["fv"].nbytes)


def test_roundtrip_random():
    rng = np.random.default_rng(0)
    bio = io.BytesIO()
    mats = {"k%d" % i: rng.standard_normal((rng.integers(1, 50), 23)).astype(np.float32) for i in range(20)}
    for k, m in mats.items():
        kaldi_io.write_mat(bio, m, key=k)
    back = dict(kaldi_io.read_mat_ark(io.BytesIO(bio.getvalue())))
    assert list(back) == list(mats)
    assert all(np.array_equal(back[k], mats[k]) for k in mats)


def test_buffered_stream_hands_back_unread_bytes(g):
    """read_mat_ark reads ahead in blocks; when the caller stops early on a seekable stream the position is restored
    to the first unread record, so interleaving with other readers keeps working."""
    bio = io.BytesIO(g["mat_ark"].tobytes())
    gen = kaldi_io.read_mat_ark(bio)
    k, m = next(gen)
    assert k == "utt-a.1"
    gen.close()
    assert kaldi_io.read_key(bio) == "utt_b/2"


def test_write_vec_flt_batch_equals_per_key_writes(g):
    a, b = io.BytesIO(), io.BytesIO()
    keys = ["k1", "k2", "k3"]
    vecs = [g["fv"], g["fv"] * 2, g["fv"][:7].copy()]
    for k, v in zip(keys, vecs):
        kaldi_io.write_vec_flt(a, v, key=k)
    kaldi_io.write_vec_flt_batch(b, keys, vecs)
    assert a.getvalue() == b.getvalue()
    with pytest.raises(kaldi_io.UnsupportedDataType):
        kaldi_io.write_vec_flt_batch(io.BytesIO(), ["x"], [np.zeros(3)])


def test_native_record_writer_writes_the_bytes_of_write_vec_flt(tmp_path, monkeypatch):
    """xv_vec_records_write_fd (csrc/xv_host.cpp) behind write_vec_flt_batch: into a TableWriter (ark + scp) and into a plain file,
    from a block with a row stride (the gathered [emitted? | x-vector] rows), with an emitted mask, keys as a list and as a
    KeyRange over an scp's text, ragged key lengths, an empty key, more records than one 4 MB piece -- byte for byte what the
    per-key write_vec_flt (the reference's framing, pinned by ark_io.npz elsewhere in this file) and the Python batch path write."""
    rng = np.random.default_rng(3)
    n, dim = 5000, 512                                                  # 10 MB of records: three pieces
    keys = ["spk%04d-u%d" % (i // 7, i * 37) for i in range(n)]          # ragged lengths
    block = rng.standard_normal((n, dim + 1)).astype(np.float32)
    emitted = rng.random(n) > 0.1
    block[:, 0] = emitted

    def python_way(path_ark, path_scp, with_mask):
        monkeypatch.setenv("XVECTOR_NATIVE_WRITER", "0")
        with kaldi_io.TableWriter(str(path_ark), str(path_scp), scp_ark_name="final.ark") as out:
            out.write(b"")
            kaldi_io.write_vec_flt(out, block[0, 1:4].copy(), key="first")   # the table already holds a record: offsets continue
            kaldi_io.write_vec_flt_batch(out, keys, block[:, 1:], emitted if with_mask else None)
            kaldi_io.write_vec_flt(out, block[1, 1:6].copy(), key="last")    # ... and per-key writes go on behind the batch
        monkeypatch.delenv("XVECTOR_NATIVE_WRITER")
        return open(path_ark, "rb").read(), open(path_scp, "rb").read()

    for with_mask in (True, False):
        want_ark, want_scp = python_way(tmp_path / "p.ark", tmp_path / "p.scp", with_mask)
        for as_range in (False, True):
            k = kaldi_io.KeyRange.from_keys(keys) if as_range else keys
            with kaldi_io.TableWriter(str(tmp_path / "n.ark"), str(tmp_path / "n.scp"), scp_ark_name="final.ark") as out:
                kaldi_io.write_vec_flt(out, block[0, 1:4].copy(), key="first")
                kaldi_io.write_vec_flt_batch(out, k, block[:, 1:], emitted if with_mask else None)
                kaldi_io.write_vec_flt(out, block[1, 1:6].copy(), key="last")
                assert out._pos == len(want_ark)
            assert open(tmp_path / "n.ark", "rb").read() == want_ark and open(tmp_path / "n.scp", "rb").read() == want_scp
    # the scp's offsets are the ones the reader seeks to
    table = dict(ln.split() for ln in want_scp.decode().splitlines())
    pos = int(table[keys[-1]].split(":")[1])
    assert np.array_equal(kaldi_io.read_vec_flt(io.BytesIO(want_ark[pos:])), block[-1, 1:])
    assert np.array_equal(kaldi_io.read_vec_flt(io.BytesIO(want_ark[int(table["last"].split(":")[1]):])), block[1, 1:6])
    # a plain file (no scp), per-key writes as the yardstick, an empty key among them
    some, vec = ["a", "", "ccc"], block[:3, 1:9]
    with open(tmp_path / "plain.ark", "wb") as f:
        kaldi_io.write_vec_flt_batch(f, some, vec)
    ref = io.BytesIO()
    for kk, v in zip(some, vec):
        kaldi_io.write_vec_flt(ref, np.ascontiguousarray(v), key=kk)
    assert open(tmp_path / "plain.ark", "rb").read() == ref.getvalue()
    # and a sink that is neither (BytesIO) still goes the Python way, mask included
    bio = io.BytesIO()
    kaldi_io.write_vec_flt_batch(bio, keys[:50], block[:50, 1:], emitted[:50])
    ref = io.BytesIO()
    for kk, v, ok in zip(keys[:50], block[:50, 1:], emitted[:50]):
        if ok:
            kaldi_io.write_vec_flt(ref, np.ascontiguousarray(v), key=kk)
    assert bio.getvalue() == ref.getvalue()


@pytest.mark.parametrize("no_host_lib", [False, True])
def test_scp_text_line_index_follows_the_python_split(tmp_path, monkeypatch, no_host_lib):
    """kaldi_io.ScpText (xv_scp_line_index): the lines / keys of a table by position -- the same lines the Python split yields
    (blank lines dropped, CRLF and non-ASCII texts handed to the Python rule), keys decoded lazily, ranges and masks as views."""
    if no_host_lib:
        monkeypatch.setenv("XVECTOR_NO_HOST_LIB", "1")
    monkeypatch.setattr(kaldi_io, "_HOST_LIB", False)
    keys = ["utt%03d-%s" % (i, "x" * (i % 5)) for i in range(101)]
    text = "".join("%s /data/f.ark:%d\n" % (k, 17 + 31 * i) + ("\n" if i % 13 == 0 else "   \n" if i % 17 == 0 else "") for i, k in enumerate(keys))
    path = tmp_path / "t.scp"
    path.write_text(text[:-1])                                          # no newline at the end of the file
    t = kaldi_io.ScpText(str(path))
    assert len(t) == 101 and t.native == (not no_host_lib)
    assert t.keys() == keys and list(t.keys(10, 20)) == keys[10:20] and len(t.keys(5, 5)) == 0
    want = [ln + "\n" for ln in text.splitlines() if ln.strip()]
    assert t.lines(0, 101) == want and t.lines(30, 61) == want[30:61] and t.lines(7, 7) == []
    mask = np.arange(101) % 3 == 0
    assert t.keys()[mask] == [k for k, m in zip(keys, mask) if m] and t.keys()[5:9] == keys[5:9] and t.keys()[100] == keys[100]
    assert kaldi_io.KeyRange.from_keys(keys)[mask] == t.keys()[mask]
    (tmp_path / "crlf.scp").write_bytes(b"a x.ark:1\r\nb x.ark:2\r\n")
    c = kaldi_io.ScpText(str(tmp_path / "crlf.scp"))
    assert not c.native and c.keys() == ["a", "b"] and c.lines(0, 2) == ["a x.ark:1\n", "b x.ark:2\n"]
    (tmp_path / "lead.scp").write_bytes(b"  a x.ark:1\nb\tx.ark:2\n")
    c = kaldi_io.ScpText(str(tmp_path / "lead.scp"))
    assert c.keys() == ["a", "b"] and [ln.split() for ln in c.lines(0, 2)] == [["a", "x.ark:1"], ["b", "x.ark:2"]]
    (tmp_path / "empty.scp").write_bytes(b"")
    assert len(kaldi_io.ScpText(str(tmp_path / "empty.scp"))) == 0
    monkeypatch.setattr(kaldi_io, "_HOST_LIB", False)


def test_pipe_and_large_stream_through_buffered_reader(tmp_path):
    rng = np.random.default_rng(1)
    mats = {"u%05d" % i: rng.standard_normal((int(rng.integers(1, 400)), 23)).astype(np.float32) for i in range(300)}
    ark = tmp_path / "big.ark"
    with open(ark, "wb") as f:
        for k, m in mats.items():
            kaldi_io.write_mat(f, m, key=k)
    got = dict(kaldi_io.read_mat_ark("ark:cat %s |" % ark))
    assert list(got) == list(mats) and all(np.array_equal(got[k], mats[k]) for k in mats)


@pytest.mark.parametrize("no_host_lib", [False, True])
def test_read_mat_ark_blocks_equals_the_per_record_reader(monkeypatch, no_host_lib):
    """The block reader (native scan + one GIL-free gather per pass) returns exactly the records of read_mat_ark, in order,
    across runs of float matrices with changing column counts, a double-precision record in the middle, empty matrices and
    a stream delivered in small pieces; without the host library it degrades to one-record blocks."""
    import kaldi_io
    if no_host_lib:
        monkeypatch.setenv("XVECTOR_NO_HOST_LIB", "1")
    monkeypatch.setattr(kaldi_io, "_HOST_LIB", False)            # re-probe under the env setting
    rng = np.random.default_rng(5)
    recs = [("a%03d" % i, rng.standard_normal((int(t), 7)).astype(np.float32)) for i, t in enumerate(rng.integers(0, 40, 300))]
    recs += [("dbl", rng.standard_normal((9, 7)))]                                               # DM record: generic path
    recs += [("b%03d" % i, rng.standard_normal((int(t), 3)).astype(np.float32)) for i, t in enumerate(rng.integers(1, 9, 50))]
    recs += [("c%03d" % i, rng.standard_normal((5, 7)).astype(np.float32)) for i in range(20)]
    bio = io.BytesIO()
    for k, m in recs:
        kaldi_io.write_mat(bio, m, key=k)
    raw = bio.getvalue()

    class Dribble(io.RawIOBase):                       # a pipe-like stream that returns at most 777 bytes per read
        def __init__(self, data):
            self.data, self.p = data, 0

        def readable(self):
            return True

        def readinto(self, b):
            n = min(len(b), 777, len(self.data) - self.p)
            b[:n] = self.data[self.p:self.p + n]
            self.p += n
            return n

    for make in (lambda: io.BytesIO(raw), lambda: io.BufferedReader(Dribble(raw), buffer_size=512)):
        got = []
        nblocks = 0
        for keys, feats, off in kaldi_io.read_mat_ark_blocks(make()):
            nblocks += 1
            assert feats.dtype == np.float32 and feats.flags["C_CONTIGUOUS"] and off[0] == 0 and off[-1] == feats.shape[0]
            got += [(k, feats[off[i]:off[i + 1]]) for i, k in enumerate(keys)]
        ref = list(kaldi_io.read_mat_ark(make()))
        assert [k for k, _ in got] == [k for k, _ in ref] == [k for k, _ in recs]
        for (_, a), (_, b), (_, m) in zip(got, ref, recs):
            assert a.shape == b.shape == m.shape and np.array_equal(a, np.asarray(b, np.float32)) and np.array_equal(a, m.astype(np.float32))
        assert nblocks == len(recs) if no_host_lib else nblocks < 20
    monkeypatch.setattr(kaldi_io, "_HOST_LIB", False)


def test_table_writer_batch_equals_per_record_writes(tmp_path):
    """write_vec_flt_batch on a TableWriter (one ark write + one scp write per batch) == one write_vec_flt per key:
    identical ark bytes, identical scp lines, and every scp offset opens on its vector."""
    import kaldi_io
    rng = np.random.default_rng(2)
    keys = ["spk%d-utt%04d" % (i % 7, i) for i in range(200)]
    vecs = [rng.standard_normal(int(d)).astype(np.float32) for d in rng.integers(1, 40, 200)]
    a1, s1, a2, s2 = (str(tmp_path / n) for n in ("a1.ark", "s1.scp", "a2.ark", "s2.scp"))
    with kaldi_io.TableWriter(a1, s1, scp_ark_name="X.ark") as w:
        kaldi_io.write_vec_flt_batch(w, keys[:120], vecs[:120])
        kaldi_io.write_vec_flt_batch(w, keys[120:], vecs[120:])
    with kaldi_io.TableWriter(a2, s2, scp_ark_name="X.ark") as w:
        for k, v in zip(keys, vecs):
            kaldi_io.write_vec_flt(w, v, key=k)
    assert open(a1, "rb").read() == open(a2, "rb").read() and open(s1).read() == open(s2).read()
    text = open(s1).read().replace("X.ark", a1)
    open(s1, "w").write(text)
    got = dict(kaldi_io.read_vec_flt_scp(s1))
    assert list(got) == keys and all(np.array_equal(got[k], v) for k, v in zip(keys, vecs))


def test_batch_vector_writer_equals_per_record_writer(tmp_path):
    """write_vec_flt_batch == a loop of write_vec_flt, byte for byte: [n, D] array or list input, keys of one or several
    lengths, ragged dimensions, plain stream and ark+scp table (scp offsets included), empty batch."""
    rng = np.random.default_rng(5)
    mat = rng.standard_normal((7, 12)).astype(np.float32)
    cases = [
        (["utt%03d" % i for i in range(7)], mat),                                        # uniform keys, 2-D array
        (["utt%03d" % i for i in range(7)], [mat[i] for i in range(7)]),                 # uniform keys, list of rows
        (["a", "bb", "ccc-1", "d", "ee", "f_f", "g.7"], mat),                            # keys of several lengths
        (["k1", "k22", "k333"], [mat[0], mat[1][:5], mat[2][:9]]),                       # ragged dimensions
        ([], np.zeros((0, 12), np.float32)),
    ]
    for n, (keys, vecs) in enumerate(cases):
        one, many = io.BytesIO(), io.BytesIO()
        for k, v in zip(keys, vecs):
            kaldi_io.write_vec_flt(one, np.asarray(v), key=k)
        kaldi_io.write_vec_flt_batch(many, keys, vecs)
        assert one.getvalue() == many.getvalue(), n
        a1, s1, a2, s2 = (str(tmp_path / ("%s%d" % (x, n))) for x in ("a1_", "s1_", "a2_", "s2_"))
        with kaldi_io.TableWriter(a1, s1, "X.ark") as t1:
            for k, v in zip(keys, vecs):
                kaldi_io.write_vec_flt(t1, np.asarray(v), key=k)
        with kaldi_io.TableWriter(a2, s2, "X.ark") as t2:
            kaldi_io.write_vec_flt_batch(t2, keys, vecs)
            kaldi_io.write_vec_flt_batch(t2, keys, vecs)                                  # offsets keep counting across batches
        assert open(a2, "rb").read() == 2 * open(a1, "rb").read()
        lines1, lines2 = open(s1).read().splitlines(), open(s2).read().splitlines()
        assert lines2[:len(lines1)] == lines1 and len(lines2) == 2 * len(lines1)
    with pytest.raises(kaldi_io.UnsupportedDataType):
        kaldi_io.write_vec_flt_batch(io.BytesIO(), ["a"], np.zeros((1, 3), np.float64))


def test_scp_table_block_reader_equals_entry_reads(tmp_path):
    """kaldi_io.MatScp.blocks(): an scp that follows its arks is read as streams (few large blocks), a subset / shuffled /
    multi-ark / compressed-record table falls back where needed -- always the same (key, matrix) sequence as read_mat_scp."""
    rng = np.random.default_rng(7)
    arks = []
    lines = []
    for a in range(2):
        ark, scp = str(tmp_path / ("f%d.ark" % a)), str(tmp_path / ("f%d.scp" % a))
        with kaldi_io.TableWriter(ark, scp) as w:
            for i in range(40):
                m = rng.standard_normal((int(rng.integers(0, 30)), 6)).astype(np.float32 if i % 7 else np.float64)
                kaldi_io.write_mat(w, m, key="a%d-utt%03d" % (a, i))
        lines += open(scp).read().splitlines(True)
    tables = {
        "in order": lines,
        "subset": [ln for i, ln in enumerate(lines) if i % 3 != 1],
        "shuffled": [lines[i] for i in rng.permutation(len(lines))],
        "tail then head": lines[50:] + lines[:50],
        "one": lines[17:18],
        "empty": [],
    }
    for name, tl in tables.items():
        path = str(tmp_path / "t.scp")
        open(path, "wt").write("".join(tl))
        want = list(kaldi_io.read_mat_scp(path))
        table = kaldi_io.MatScp(path)
        assert len(table) == len(tl)
        got, nblocks = [], 0
        for keys, feats, off in table.blocks():
            nblocks += 1
            assert feats.dtype == np.float32 and off[0] == 0 and len(off) == len(keys) + 1
            got += [(k, feats[off[i]:off[i + 1]]) for i, k in enumerate(keys)]
        assert [k for k, _ in got] == [k for k, _ in want], name
        assert all(np.array_equal(g, np.asarray(w, np.float32)) for (_, g), (_, w) in zip(got, want)), name
        assert [k for k, _ in table] == [k for k, _ in want]
        if name == "in order":
            assert nblocks <= 30          # float64 records split the runs; still far fewer blocks than utterances


def test_vector_block_reader_and_vector_scp_table(tmp_path):
    """read_vec_flt_ark_blocks / VecScp.blocks(): a VAD-like ark (float vectors of various lengths, one double-precision
    record, one empty vector) comes out in scanner passes with the same keys and values as read_vec_flt_ark / _scp."""
    rng = np.random.default_rng(9)
    ark, scp = str(tmp_path / "vad.ark"), str(tmp_path / "vad.scp")
    with kaldi_io.TableWriter(ark, scp) as w:
        for i in range(60):
            v = (rng.random(int(rng.integers(0, 50))) > 0.3).astype(np.float64 if i == 31 else np.float32)
            kaldi_io.write_vec_flt(w, v, key="utt%02d" % i)
    want = list(kaldi_io.read_vec_flt_ark(ark))
    got, nblocks = [], 0
    for keys, vals, off in kaldi_io.read_vec_flt_ark_blocks(ark):
        nblocks += 1
        assert vals.ndim == 1 and vals.dtype == np.float32
        got += [(k, vals[off[i]:off[i + 1]]) for i, k in enumerate(keys)]
    assert [k for k, _ in got] == [k for k, _ in want] and nblocks <= 4
    assert all(np.array_equal(g, np.asarray(w_, np.float32)) for (_, g), (_, w_) in zip(got, want))
    table = kaldi_io.VecScp(scp)
    got2 = [(k, vals[off[i]:off[i + 1]]) for keys, vals, off in table.blocks() for i, k in enumerate(keys)]
    assert [k for k, _ in got2] == [k for k, _ in want]
    assert all(np.array_equal(g, np.asarray(w_, np.float32)) for (_, g), (_, w_) in zip(got2, want))
    assert [(k, v.tolist()) for k, v in table] == [(k, np.asarray(v, np.float32).tolist()) for k, v in kaldi_io.read_vec_flt_scp(scp)]


def test_in_place_windows_equal_the_record_reader(tmp_path):
    """scan_mat_ark_windows / MatScp.windows: the matrices are used where the stream was read to (ArkArena), nothing is
    gathered -- every arena size (records cut by the arena's end are carried over, a record larger than an arena and a
    double-precision record go through the generic reader, a column-count change inside an arena splits the item), the short
    first fill, a truncated stream, early close handing the unread bytes back, and an scp table with a gap in it."""
    import io
    if kaldi_io._host_lib() is None:
        pytest.skip("host library not built")
    rng = np.random.default_rng(0)
    bio = io.BytesIO()
    mats = []
    for i in range(300):
        T, F = int(rng.integers(0, 60)), (23 if i < 250 else 5)
        m = rng.standard_normal((T, F)).astype(np.float32)
        kaldi_io.write_mat(bio, m.astype(np.float64) if i == 100 else m, key="k%d" % i)
        mats.append(m)
    data = bio.getvalue()
    for cap, first in ((1 << 20, None), (20000, None), (9000, 3000), (700, None), (6000, 100)):
        arenas = []

        def take():
            arenas.append(kaldi_io.ArkArena(cap))
            return arenas[-1]
        got, holders = [], []
        for keys, addr, rows, cols, holder in kaldi_io.scan_mat_ark_windows(io.BytesIO(data), take, first):
            am = kaldi_io.ArkMats()
            am.add(addr, rows, cols, holder)
            assert len(am) == len(keys) and am.uniform_cols() == cols and am.frames() == int(np.sum(rows))
            got += [(k, np.array(am[j])) for j, k in enumerate(keys)]
            if isinstance(holder, kaldi_io.ArkArena):
                assert all(holder is not h for h in holders)       # an arena is handed out with exactly one item
                holders.append(holder)
        assert [k for k, _ in got] == ["k%d" % i for i in range(300)], cap
        assert all(a.shape == b.shape and np.array_equal(a.astype(np.float32), b) for (_, a), b in zip(got, mats))
    # a plain BytesIO is copied by the host library (no interpreter lock), any other stream through its readinto: same records,
    # and the BytesIO is free again afterwards (its exported buffer is released: it can be written to and closed)
    class Stream(io.BytesIO):
        pass
    for make in (io.BytesIO, Stream):
        src = make(data)
        keys2 = [k for item in kaldi_io.scan_mat_ark_windows(src, lambda: kaldi_io.ArkArena(20000)) for k in item[0]]
        assert keys2 == ["k%d" % i for i in range(300)]
        src.write(b"x")
        src.close()
    with pytest.raises(Exception):
        list(kaldi_io.scan_mat_ark_windows(io.BytesIO(data[:-7]), lambda: kaldi_io.ArkArena(1 << 20)))
    f = io.BytesIO(data)
    g = kaldi_io.scan_mat_ark_windows(f, lambda: kaldi_io.ArkArena(20000))
    first_keys = next(g)[0]
    g.close()
    rest = list(kaldi_io.read_mat_ark(f))
    assert rest[0][0] == "k%d" % len(first_keys) and len(first_keys) + len(rest) == 300
    # scp table: in-order run, one entry skipped (the run ends there and restarts), everything verified against the keys
    ark, scp = str(tmp_path / "f.ark"), str(tmp_path / "f.scp")
    with kaldi_io.TableWriter(ark, scp) as tw:
        for i in range(120):
            kaldi_io.write_mat(tw, mats[i] if i != 100 else mats[i].astype(np.float64), key="k%d" % i)
    lines = open(scp).read().splitlines()
    sub = str(tmp_path / "sub.scp")
    open(sub, "wt").write("\n".join(l for i, l in enumerate(lines) if i != 57) + "\n")
    got = []
    for keys, addr, rows, cols, holder in kaldi_io.MatScp(sub).windows(lambda: kaldi_io.ArkArena(30000)):
        am = kaldi_io.ArkMats()
        am.add(addr, rows, cols, holder)
        got += [(k, np.array(am[j])) for j, k in enumerate(keys)]
    want = [i for i in range(120) if i != 57]
    assert [k for k, _ in got] == ["k%d" % i for i in want]
    assert all(np.array_equal(a.astype(np.float32), mats[i]) for (_, a), i in zip(got, want))


def test_compressed_matrices_through_the_in_place_reader(tmp_path):
    """Kaldi's default feature arks are CompressedMatrix records ("CM "): the in-place reader hands them to the host library's
    decoder (xv_ark_decode_cm: Kaldi's float32 arithmetic, a few threads, into arenas of their own) instead of reading them one
    by one through NumPy.  Bit-identical to the generic reader (which the reference-recorded fixtures pin) on a stream that mixes
    compressed and plain matrices, widths that change, a one-row matrix and a matrix larger than an arena; input order kept;
    every arena accounted for; the same through an scp table."""
    import io
    from fixture_inputs import encode_cm_record
    if kaldi_io._host_lib() is None or not hasattr(kaldi_io._host_lib(), "xv_ark_decode_cm"):
        pytest.skip("host library not built")
    rng = np.random.default_rng(11)
    bio, want = io.BytesIO(), []
    for i in range(400):
        t = 1 if i == 7 else 3000 if i == 150 else int(rng.integers(2, 300))
        f = 40 if 200 <= i < 260 else 23
        m = (rng.standard_normal((t, f)) * (1 + i % 4) + (i % 3)).astype(np.float32)
        if i % 9 == 4:
            kaldi_io.write_mat(bio, m, key="plain%03d" % i)
        elif i % 31 == 5:                                          # the two-byte / one-byte forms (Kaldi: matrices of <= 8 rows)
            import struct
            two = i % 2 == 0
            u = rng.integers(0, 65536 if two else 256, size=m.shape).astype("<u2" if two else np.uint8)
            bio.write(("utt%03d" % i).encode() + (b" \0BCM2 " if two else b" \0BCM3 ") + struct.pack("<ffii", -3.0, 7.5, t, f) + u.tobytes())
        else:
            bio.write(encode_cm_record("utt%03d" % i, m))
        want.append(("plain%03d" % i) if i % 9 == 4 else ("utt%03d" % i))
    raw = bio.getvalue()
    ref = dict(kaldi_io.read_mat_ark(io.BytesIO(raw)))              # the generic reader: NumPy decode, record by record
    assert list(ref) == want
    taken, released, held, got = [], [], [], {}

    def take():
        taken.append(kaldi_io.ArkArena(1 << 18))                    # 256 KB: the 3000 x 23 matrix (276 KB decoded) does not fit
        return taken[-1]
    order = []
    for keys, addr, rows, cols, holder in kaldi_io.scan_mat_ark_windows(io.BytesIO(raw), take, None, released.append):
        am = kaldi_io.ArkMats()
        am.add(addr, rows, cols, holder)
        for j, k in enumerate(keys):
            got[k] = np.array(am[j])
        order += keys
        if isinstance(holder, kaldi_io.ArkArena):
            held.append(holder)
    assert order == want
    assert all(got[k].dtype == np.float32 and np.array_equal(got[k], ref[k]) for k in want)
    assert sorted(map(id, taken)) == sorted(map(id, held + released)) and len(set(map(id, taken))) == len(taken)
    # a consumer that recycles every holder at once, from a small pool: an arena that went downstream as the holder of a plain
    # matrix must not come back as a decode arena while its own tail is still being scanned
    pool = [kaldi_io.ArkArena(1 << 18) for _ in range(4)]
    free, again = list(pool), {}
    for keys, addr, rows, cols, holder in kaldi_io.scan_mat_ark_windows(io.BytesIO(raw), free.pop, 4096, free.append):
        am = kaldi_io.ArkMats()
        am.add(addr, rows, cols, holder)
        for j, k in enumerate(keys):
            again[k] = np.array(am[j])
        if isinstance(holder, kaldi_io.ArkArena):
            free.append(holder)
    assert list(again) == want and all(np.array_equal(again[k], ref[k]) for k in want) and len(free) == 4
    # the same records behind an scp table
    ark, scp = str(tmp_path / "c.ark"), str(tmp_path / "c.scp")
    lines, pos = [], 0
    with open(ark, "wb") as f:
        f.write(raw)
    stream = io.BytesIO(raw)
    while True:
        at = stream.tell()
        key = kaldi_io.read_key(stream)
        if not key:
            break
        lines.append("%s %s:%d" % (key, ark, at + len(key) + 1))
        kaldi_io.read_mat(stream)
    open(scp, "wt").write("\n".join(lines) + "\n")
    n = 0
    for keys, addr, rows, cols, holder in kaldi_io.MatScp(scp).windows(lambda: kaldi_io.ArkArena(1 << 18), None, lambda a: None):
        am = kaldi_io.ArkMats()
        am.add(addr, rows, cols, holder)
        for j, k in enumerate(keys):
            assert np.array_equal(np.array(am[j]), ref[k])
            n += 1
    assert n == len(want)


def test_a_compressed_record_the_decoder_refuses_raises_instead_of_spinning():
    """"CM " header with rows = -1.  The native decoder stops at it with nothing decoded; the in-place reader used
    to go back to the float-matrix scanner, which stopped at the same byte -- for ever, one arena per turn.  Now the record goes to the
    generic reader, which raises as it does for any malformed matrix; good records in front of it still come out."""
    import io, struct
    from fixture_inputs import encode_cm_record
    if kaldi_io._host_lib() is None or not hasattr(kaldi_io._host_lib(), "xv_ark_decode_cm"):
        pytest.skip("host library not built")
    rng = np.random.default_rng(3)
    good = encode_cm_record("good", rng.standard_normal((12, 5)).astype(np.float32))
    bad = b"utt1 \0BCM " + struct.pack("<ffii", -1.0, 2.0, -1, 5) + bytes(64)
    for raw, n_good in ((bad, 0), (good + bad, 1), (good + good + bad + good, 2)):
        taken = []

        def take():
            assert len(taken) < 20, "the reader keeps taking arenas at a record it cannot decode"
            taken.append(kaldi_io.ArkArena(1 << 16))
            return taken[-1]
        keys = []
        with pytest.raises(kaldi_io.BadInputFormat):
            for item in kaldi_io.scan_mat_ark_windows(io.BytesIO(raw), take, None, lambda a: None):
                keys += item[0]
        assert keys == ["good"] * n_good
        with pytest.raises(kaldi_io.BadInputFormat):                  # the generic reader on the same bytes
            list(kaldi_io.read_mat_ark(io.BytesIO(raw)))


def test_subset_scp_tables_are_read_in_place(tmp_path):
    """A feats.scp that lists a SUBSET of its ark, in the ark's order (utterances removed by a filter -- the usual state of a Kaldi
    data directory): the in-place reader skips the records the table leaves out instead of giving up at the first gap and reading
    entry by entry.  Same matrices as the record-by-record reader; nearly all of them come out of arenas; a sparse table (one entry
    in ten) and a table in another order than its ark still read correctly (entry by entry)."""
    if kaldi_io._host_lib() is None:
        pytest.skip("host library not built")
    rng = np.random.default_rng(21)
    mats = [rng.standard_normal((int(rng.integers(1, 60)), 13)).astype(np.float32) for _ in range(600)]
    ark, scp = str(tmp_path / "f.ark"), str(tmp_path / "f.scp")
    with kaldi_io.TableWriter(ark, scp) as tw:
        for i, m in enumerate(mats):
            kaldi_io.write_mat(tw, m, key="utt%04d" % i)
    lines = open(scp).read().splitlines()

    def read(sel):
        p = str(tmp_path / "sel.scp")
        open(p, "wt").write("\n".join(sel) + "\n")
        out, in_place = [], 0
        for keys, addr, rows, cols, holder in kaldi_io.MatScp(p).windows(lambda: kaldi_io.ArkArena(1 << 16), None, lambda a: None):
            am = kaldi_io.ArkMats()
            am.add(addr, rows, cols, holder)
            out += [(k, np.array(am[j])) for j, k in enumerate(keys)]
            in_place += len(keys) if isinstance(holder, (kaldi_io.ArkArena, list)) else 0
        return out, in_place
    for name, sel in (("90 %", [l for l in lines if rng.random() < 0.9]), ("50 %", [l for l in lines if rng.random() < 0.5]),
                      ("one in ten", lines[::10]), ("reversed", lines[::-1][:80])):
        got, in_place = read(sel)
        idx = [int(l.split()[0][3:]) for l in sel]
        assert [k for k, _ in got] == [l.split()[0] for l in sel], name
        assert all(np.array_equal(a, mats[i]) for (_, a), i in zip(got, idx)), name
        if name in ("90 %", "50 %"):
            assert in_place >= 0.9 * len(sel), (name, in_place, len(sel))
        # the block form (what a vad.scp restricted to the features' keys goes through)
        blk = [(k, v[int(o[j]):int(o[j + 1])]) for ks, v, o in kaldi_io.MatScp(str(tmp_path / "sel.scp")).blocks() for j, k in enumerate(ks)]
        assert [k for k, _ in blk] == [l.split()[0] for l in sel] and all(np.array_equal(a, mats[i]) for (_, a), i in zip(blk, idx)), name


def test_every_arena_taken_comes_back(tmp_path):
    """The in-place readers account for every arena they take: it is either the holder of exactly one item (the consumer
    recycles it) or handed to ``release`` -- an arena that held only the carried bytes of an oversized record, and the arena of
    an scp run whose first key does not match the table.  With a bounded pool a lost arena is a reader that blocks forever."""
    import io
    if kaldi_io._host_lib() is None:
        pytest.skip("host library not built")
    rng = np.random.default_rng(5)
    mats = [rng.standard_normal((int(rng.integers(1, 40)), 23)).astype(np.float32) for _ in range(80)]
    mats[10] = rng.standard_normal((400, 23)).astype(np.float32)            # larger than an arena: generic reader, carried bytes
    bio = io.BytesIO()
    for i, m in enumerate(mats):
        kaldi_io.write_mat(bio, m, key="k%d" % i)
    taken, released, held = [], [], []

    def take():
        taken.append(kaldi_io.ArkArena(9000))
        return taken[-1]
    keys = []
    for k, addr, rows, cols, holder in kaldi_io.scan_mat_ark_windows(io.BytesIO(bio.getvalue()), take, None, released.append):
        keys += k
        if isinstance(holder, kaldi_io.ArkArena):
            held.append(holder)
    assert keys == ["k%d" % i for i in range(80)]
    assert sorted(map(id, taken)) == sorted(map(id, held + released)) and len(set(map(id, taken))) == len(taken)
    # scp whose keys alternate between the ark's and renamed ones: every miss gives its arena back, and after two short runs
    # the table is read entry by entry (no further arena is taken)
    ark, scp = str(tmp_path / "f.ark"), str(tmp_path / "f.scp")
    with kaldi_io.TableWriter(ark, scp) as tw:
        for i, m in enumerate(mats):
            kaldi_io.write_mat(tw, m, key="k%d" % i)
    lines = open(scp).read().splitlines()
    renamed = str(tmp_path / "renamed.scp")
    open(renamed, "wt").write("\n".join(("x" + l) if i % 2 else l for i, l in enumerate(lines)) + "\n")
    del taken[:], released[:], held[:]
    got = []
    for k, addr, rows, cols, holder in kaldi_io.MatScp(renamed).windows(take, None, released.append):
        am = kaldi_io.ArkMats()
        am.add(addr, rows, cols, holder)
        got += [(kk, np.array(am[j])) for j, kk in enumerate(k)]
        if isinstance(holder, kaldi_io.ArkArena):
            held.append(holder)
    assert [k for k, _ in got] == [("x" if i % 2 else "") + "k%d" % i for i in range(80)]
    assert all(np.array_equal(a, m) for (_, a), m in zip(got, mats))
    assert sorted(map(id, taken)) == sorted(map(id, held + released))
    assert len(taken) <= 6, "a table that does not follow its ark must stop taking arenas (took %d)" % len(taken)


def test_header_only_index_of_an_ark_file(tmp_path):
    """kaldi_io.index_mat_ark_file: offsets / shapes / keys of every record from one small pread per record (the matrices are
    hopped over) == what the record reader sees; a file with another record type or a truncated tail gives None (callers read
    the stream the ordinary way); FileRange reads exactly its bytes and leaves the file object's own position alone."""
    if kaldi_io._host_lib() is None or not hasattr(kaldi_io._host_lib(), "xv_ark_index_fd"):
        pytest.skip("host library not built")
    rng = np.random.default_rng(11)
    mats = [rng.standard_normal((int(rng.integers(0, 50)), 7)).astype(np.float32) for _ in range(200)]
    keys = ["spk%d-utt_%03d.a" % (i % 7, i) for i in range(200)]
    keys[17] = "k" * 400                                                   # a key longer than the first header read
    path = str(tmp_path / "f.ark")
    with open(path, "wb") as f:
        for k, m in zip(keys, mats):
            kaldi_io.write_mat(f, m, key=k)
    with open(path, "rb") as f:
        off, rows, cols, got_keys = kaldi_io.index_mat_ark_file(f)
        assert f.tell() == 0
        assert got_keys == keys and rows.tolist() == [m.shape[0] for m in mats] and set(cols.tolist()) == {7}
        assert off[0] == 0 and off[-1] == os.path.getsize(path) and len(off) == 201
        # a range of records read through FileRange == the same records of the whole file
        part = kaldi_io.FileRange(f, off[40], off[90])
        sub = list(kaldi_io.read_mat_ark(part))
        assert [k for k, _ in sub] == keys[40:90] and all(np.array_equal(a, b) for (_, a), b in zip(sub, mats[40:90]))
        assert f.tell() == 0
        f.seek(off[150])
        off2, _, _, keys2 = kaldi_io.index_mat_ark_file(f)                  # from the stream position on
        assert keys2 == keys[150:] and off2[0] == off[150]
        assert kaldi_io.index_mat_ark_file(f, with_keys=False)[3] is None
    # compressed speech-feature matrices are indexed the same way (their header carries the shape)
    from fixture_inputs import encode_cm_record
    cpath = str(tmp_path / "c.ark")
    cm = [rng.standard_normal((int(rng.integers(1, 300)), 23)).astype(np.float32) for _ in range(60)]
    with open(cpath, "wb") as f:
        for i, m in enumerate(cm):
            if i % 5 == 2:
                kaldi_io.write_mat(f, m, key="plain%d" % i)
            else:
                f.write(encode_cm_record("cm%d" % i, m))
    with open(cpath, "rb") as f:
        coff, crows, ccols, ckeys = kaldi_io.index_mat_ark_file(f)
        assert ckeys == [("plain%d" if i % 5 == 2 else "cm%d") % i for i in range(60)]
        assert crows.tolist() == [m.shape[0] for m in cm] and set(ccols.tolist()) == {23} and coff[-1] == os.path.getsize(cpath)
        whole = list(kaldi_io.read_mat_ark(cpath))
        sub = list(kaldi_io.read_mat_ark(kaldi_io.FileRange(f, coff[13], coff[41])))
        assert [k for k, _ in sub] == ckeys[13:41] and all(np.array_equal(a, b) for (_, a), (_, b) in zip(sub, whole[13:41]))
    # a short two-byte record ("CM2 ": what Kaldi writes for matrices of < 8 rows; r c elements, NO per-column headers) as the LAST
    # record of the file: fewer than 8 c bytes are left behind its header, and it is a complete record all the same
    import struct
    tpath = str(tmp_path / "tail.ark")
    with open(tpath, "wb") as f:
        f.write(encode_cm_record("cm0", cm[0]))
        f.write(b"short \0BCM2 " + struct.pack("<ffii", -3.0, 7.5, 2, 23) + rng.integers(0, 65536, size=(2, 23)).astype("<u2").tobytes())
    with open(tpath, "rb") as f:
        idx = kaldi_io.index_mat_ark_file(f)
        assert idx is not None and idx[3] == ["cm0", "short"] and idx[1].tolist() == [cm[0].shape[0], 2] and idx[0][-1] == os.path.getsize(tpath)
    with open(path, "ab") as f:
        kaldi_io.write_mat(f, mats[3].astype(np.float64), key="double")
    with open(path, "rb") as f:
        assert kaldi_io.index_mat_ark_file(f) is None
    cut = str(tmp_path / "cut.ark")
    open(cut, "wb").write(open(path, "rb").read()[:int(off[120]) + 9])
    with open(cut, "rb") as f:
        assert kaldi_io.index_mat_ark_file(f) is None
    import io
    assert kaldi_io.index_mat_ark_file(io.BytesIO(b"abc")) is None and not kaldi_io.is_regular_file(io.BytesIO(b""))


def test_mapped_ark_windows_equal_the_record_reader(tmp_path):
    """map_stream + scan_mat_ark_mapped: an ark that is already in memory (a BytesIO) is scanned where it lies -- same keys and
    matrices as the record reader, whatever the window size; a record the native scanner does not take (a double matrix) and
    a truncated tail go through the generic reader; files and pipes are not mapped (they are read into the arenas); the
    BytesIO stays usable (nothing of it is exported)."""
    if kaldi_io._host_lib() is None:
        pytest.skip("host library not built")
    rng = np.random.default_rng(1)
    bio = io.BytesIO()
    mats = []
    for i in range(300):
        T, F = int(rng.integers(0, 60)), (23 if i < 250 else 5)
        m = rng.standard_normal((T, F)).astype(np.float32)
        kaldi_io.write_mat(bio, m.astype(np.float64) if i == 100 else m, key="k%d" % i)
        mats.append(m)
    data = bio.getvalue()
    path = str(tmp_path / "f.ark")
    open(path, "wb").write(data)

    def collect(arr, window, first):
        got = []
        for keys, addr, rows, cols, holder in kaldi_io.scan_mat_ark_mapped(arr, window, first):
            am = kaldi_io.ArkMats()
            am.add(addr, rows, cols, holder)
            assert len(am) == len(keys) and am.uniform_cols() == cols
            got += [(k, np.array(am[j])) for j, k in enumerate(keys)]
        return got

    for window, first in ((1 << 20, None), (20000, 3000), (700, None), (64, None)):
        for src in (io.BytesIO(data),):
            arr = kaldi_io.map_stream(src)
            assert arr is not None and arr.shape[0] == len(data) and not arr.flags.writeable
            got = collect(arr, window, first)
            assert [k for k, _ in got] == ["k%d" % i for i in range(300)], (window, type(src))
            assert all(a.shape == b.shape and np.array_equal(a.astype(np.float32), b) for (_, a), b in zip(got, mats))
            assert src.read(1) == b""                                # the stream is consumed
            src.write(b"x")                                          # (no export pins the stream's buffer)
            del arr, got
            src.close()
    # from the current position only
    f = io.BytesIO(data)
    first_key, first_mat = next(iter(kaldi_io.read_mat_ark(f)))      # (the record reader hands back its read-ahead)
    assert first_key == "k0"
    rest = collect(kaldi_io.map_stream(f), 5000, None)
    assert [k for k, _ in rest] == ["k%d" % i for i in range(1, 300)]
    # truncated tail: everything before it comes out, then the generic reader raises
    with pytest.raises(Exception):
        collect(kaldi_io.map_stream(io.BytesIO(data[:-7])), 1 << 20, None)
    # not mapped: a regular file, a pipe
    with open(path, "rb") as f:
        assert kaldi_io.map_stream(f) is None and f.tell() == 0
    fd = kaldi_io.open_or_fd("cat %s |" % path)
    assert kaldi_io.map_stream(fd) is None
    assert len(fd.read()) == len(data)
    # compressed matrices behind plain ones: the mapped walk ends at the first of them and the rest goes through the arena reader
    # WITHOUT being copied first (kaldi_io.MemStream over the mapped bytes)
    from fixture_inputs import encode_cm_record
    mixed = io.BytesIO()
    for i in range(40):
        m = rng.standard_normal((int(rng.integers(1, 50)), 23)).astype(np.float32)
        if i < 15:
            kaldi_io.write_mat(mixed, m, key="m%d" % i)
        else:
            mixed.write(encode_cm_record("m%d" % i, m))
    want = list(kaldi_io.read_mat_ark(io.BytesIO(mixed.getvalue())))
    got = collect(kaldi_io.map_stream(io.BytesIO(mixed.getvalue())), 1 << 20, None)
    assert [k for k, _ in got] == [k for k, _ in want] and all(np.array_equal(a, b) for (_, a), (_, b) in zip(got, want))
    ms = kaldi_io.MemStream(np.frombuffer(b"0123456789", np.uint8))
    buf = bytearray(4)
    assert ms.read(3) == b"012" and ms.readinto(buf) == 4 and bytes(buf) == b"3456" and ms.tell() == 7
    assert ms.seek(-2, 2) == 8 and ms.read() == b"89" and ms.read(5) == b"" and ms.seek(-1, 1) == 9
    fd.close()
