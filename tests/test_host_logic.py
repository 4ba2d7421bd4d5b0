"""Host-side logic of the path (CPU-only): chunk planning, ragged layout, model-dir contract, CLI
plumbing of the extract_embedding twin, topology constants."""
import io
import json
import os

import numpy as np
import pytest

import kaldi_io
from xvector_amd import engine, synthetic, topology, weights as wio


def test_topology_flop_constants_match_survey():
    topo = topology.get("ModelWithoutDropout")
    assert topology.flops_per_frame(topo, 23) == 8506368          # 8,506,368 FLOP/frame
    assert topology.flops_per_utt(topo) == 3145728                # 3,145,728 FLOP/utt
    assert topology.flops_per_frame(topology.get("ModelWithoutDropoutTdnn"), 23) == 5360640
    assert topology.max_halo(topo) == 3 and topology.max_halo(topology.get("ModelWithoutDropoutTdnn")) == 3
    att = topology.get("ModelL2LossWithoutDropoutLReluAttention")        # models.py:992: last layer 6*512, split in two
    assert topology.pooled_dim(att) == 3072 == topology.pooled_dim(topo)
    assert topology.flops_per_frame(att, 23) == 8506368 + 2 * 512 * 1536 + 2 * 1536 * 1536
    assert topology.flops_per_utt(att) == 3145728
    with pytest.raises(KeyError):
        topology.get("NoSuchModel")


def test_plan_chunks_semantics():
    pc = engine.plan_chunks
    assert pc(0, 100, -1) is None and pc(99, 100, -1) is None
    assert pc(100, 100, -1) == [(0, 100)]
    assert pc(20024, 25, 10000) == [(0, 10000), (10000, 10000)]              # 24-frame tail dropped
    assert pc(20025, 25, 10000) == [(0, 10000), (10000, 10000), (20000, 25)]
    assert pc(500, 25, 10000) == [(0, 500)]                                  # T < chunk -> one pass


def test_batch_layout_gap_rows():
    lay = engine.BatchLayout([5, 1, 7], 3)
    assert list(lay.row_start) == [3, 11, 15] and lay.rows == 3 + 8 + 4 + 10
    m = lay.row_valid()
    assert m.sum() == 13 and m[:3].sum() == 0 and m[8:11].sum() == 0 and m[-3:].sum() == 0
    mats = [np.full((n, 2), i + 1, np.float32) for i, n in enumerate([5, 1, 7])]
    out = np.full((lay.rows + 4, 2), 9, np.float32)
    lay.pack(mats, out)
    assert (out[:lay.rows][m == 0] == 0).all() and (out[11] == 2).all() and (out[lay.rows:] == 9).all()
    empty = engine.BatchLayout([], 3)
    assert empty.rows == 3 and empty.nchunks == 0


def test_model_dir_contract_roundtrip(tmp_path):
    import models
    mdir = str(tmp_path / "nnet" / "model_0")
    m = models.ModelWithoutDropoutTdnn()
    m.build_model(64, 23, mdir, None)
    # the files the reference's drivers test for (ze_utils.py:561-567, extract_embedding.py:88)
    assert os.path.getsize(os.path.join(mdir, "model.meta")) > 0 and open(os.path.join(mdir, "done")).read() == "done"
    assert wio.is_correct_model_dir(mdir)
    meta = json.load(open(os.path.join(mdir, "model.meta")))
    assert meta["model_class"] == "ModelWithoutDropoutTdnn" and meta["topology"]["dilations"] == [1, 2, 3, 1, 1]
    w = m.get_models_weights(mdir)
    assert w["frame_level_info_layer-1/w:0"].shape == (3, 512, 512)        # [K, Cin, Cout] as TF stores it
    assert w["embed_layer-0/w:0"].shape == (3072, 512) and w["output/w:0"].shape == (512, 64)
    assert np.all(w["frame_level_info_layer-0/b:0"] == np.float32(0.1))
    assert np.all(w["frame_level_info_layer-3/variance:0"] == 1) and np.all(w["embed_layer-1/gamma:0"] == 1)
    assert sorted(w) == sorted(wio.expected_names(meta["topology"]))
    assert w["output/w:0"].flags.writeable                                  # own arrays, not views of the weight file
    # model.h5 (models.py:180-214): written and then served from when h5py exists -- exercised with a minimal stand-in here
    assert not os.path.exists(os.path.join(mdir, "model.h5"))               # no h5py in this image: nothing written
    import sys, types

    class _Group(object):
        pass

    class _Data(object):
        def __init__(self, a): self.a = a
        def __getitem__(self, k): return self.a

    class _File(object):
        store = {}
        def __init__(self, path, mode):
            self.path, self.mode = path, mode
            if mode == "w":
                open(path, "wb").write(b"h5-stand-in")
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def create_dataset(self, name, data): _File.store.setdefault(self.path, {})[name] = np.array(data)
        def visititems(self, fn):
            for k, v in _File.store[self.path].items():
                fn(k, _Data(v))
    fake = types.ModuleType("h5py")
    fake.Group, fake.File = _Group, _File
    sys.modules["h5py"] = fake
    try:
        first = m.get_models_weights(mdir)
        assert os.path.exists(os.path.join(mdir, "model.h5")) and sorted(_File.store[os.path.join(mdir, "model.h5")]) == sorted(w)
        os.remove(os.path.join(mdir, "model.weights.npz"))                  # second call must come from the h5 alone
        again = m.get_models_weights(mdir)
        assert sorted(again) == sorted(first) and all(np.array_equal(again[k], w[k]) and again[k].dtype == np.float32 for k in w)
    finally:
        del sys.modules["h5py"]
    # incomplete dirs are rejected
    os.remove(os.path.join(mdir, "done"))
    assert not wio.is_correct_model_dir(mdir)
    tfdir = tmp_path / "tf_ckpt"
    tfdir.mkdir()
    (tfdir / "model.meta").write_bytes(b"\x0a\x10not-json-protobuf")
    with pytest.raises(IOError):
        wio.load_model_dir(str(tfdir))


def test_prelu_model_has_alpha_variables(tmp_path):
    import models
    mdir = str(tmp_path / "m")
    models.ModelWithoutDropoutPRelu().build_model(8, 23, mdir, None)
    w, meta = wio.load_model_dir(mdir)
    assert np.all(w["frame_level_info_layer-2/prelu/prelu:0"] == np.float32(0.1))     # tf_block.py:45-46
    assert w["embed_layer-0/prelu/prelu:0"].shape == (512,)


def test_class_selection_by_name_like_train_dnn():
    import models
    for name in ("Model", "ModelWithoutDropout", "ModelWithoutDropoutTdnn", "ModelWithoutDropoutPRelu",
                 "ModelL2LossWithoutDropoutPRelu", "ModelL2LossWithoutDropoutLRelu", "ModelL2LossWithoutDropoutReluHeInit"):
        obj = eval("models.%s()" % name)                                   # train_dnn.py:492
        assert isinstance(obj, models.Model) and obj.class_topology()["layer_sizes"][-1] == 1536
    assert models.Model().create_one_hot_output_matrix.__self__ is not None


def test_process_wspecifier_and_args(tmp_path):
    import extract_embedding as ee
    w, ark, scp = ee.process_wspecifier("ark:| copy-vector ark:- ark,scp:/x/xv.1.ark,/x/xv.1.scp")
    assert (ark, scp) == ("/x/xv.1.ark", "/x/xv.1.scp") and w.endswith("ark,scp:/x/xv.1.ark.tmp.ark,/x/xv.1.scp.tmp.scp")
    w, ark, scp = ee.process_wspecifier("scp,ark:/x/a.scp,/x/a.ark")
    assert (ark, scp) == ("/x/a.ark", "/x/a.scp") and w == "scp,ark:/x/a.scp.tmp.scp,/x/a.ark.tmp.ark"
    assert ee.process_wspecifier("ark:/x/plain.ark") == ("ark:/x/plain.ark", None, None)
    with pytest.raises(Exception):
        ee.get_args(["--feature-rspecifier", "ark:a", "--vector-wspecifier", "ark:b", "--model-dir", str(tmp_path)])
    (tmp_path / "model.meta").write_text("{}")
    a = ee.get_args(["--feature-rspecifier", "ark:a", "--vector-wspecifier", "ark:b", "--model-dir", str(tmp_path)])
    assert a.min_chunk_size == 100 and a.chunk_size == -1 and a.use_gpu == "no"       # reference defaults


def test_extract_embedding_skips_when_outputs_exist(tmp_path):
    import extract_embedding as ee
    (tmp_path / "model.meta").write_text("{}")
    ark, scp = tmp_path / "o.ark", tmp_path / "o.scp"
    ark.write_bytes(b"x"); scp.write_text("x")
    # returns before touching the (invalid) model or the (missing) input: extract_embedding.py:126-128
    ee.main(["--feature-rspecifier", "ark:/nonexistent", "--vector-wspecifier", "ark,scp:%s,%s" % (ark, scp),
             "--model-dir", str(tmp_path)])


def test_extract_embedding_exits_1_on_error(tmp_path):
    import extract_embedding as ee
    (tmp_path / "model.meta").write_text("{}")
    with pytest.raises(SystemExit) as e:
        ee.main(["--feature-rspecifier", "ark:/nonexistent.ark", "--vector-wspecifier", "ark:%s" % (tmp_path / "o.ark"),
                 "--model-dir", str(tmp_path)])
    assert e.value.code == 1


def test_synthetic_generators_are_deterministic():
    a = synthetic.utterance_lengths(100, 200, 400, 1234)
    assert a.min() >= 200 and a.max() <= 400 and np.array_equal(a, synthetic.utterance_lengths(100, 200, 400, 1234))
    u = synthetic.make_utterances(3, 25, 30, 23, 1234)
    assert u[0][0] == "utt000000" and u[0][1].dtype == np.float32 and u[0][1].shape[1] == 23
    w1 = synthetic.trained_like(synthetic.SMALL_TOPOLOGY, 5, seed=3)
    w2 = synthetic.trained_like(synthetic.SMALL_TOPOLOGY, 5, seed=3)
    assert all(np.array_equal(w1[k], w2[k]) for k in w1)


def test_schedules_match_reference_goldens():
    """local/tf/ze_utils.py twin: learning-rate and dropout schedules == values recorded from the reference's own
    functions (tests/golden/make_golden.py schedules)."""
    import ze_utils
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "schedules.npz")) as g:
        for a, v in zip(g["lr_args"], g["lr_vals"]):
            got = ze_utils.get_learning_rate(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]), float(a[6]))
            assert got == v, (a, got, v)                      # same formula, same float64 op order -> identical
        for sc, row in zip(g["schedules"], g["dropout_table"]):
            for f, v in zip(g["fractions"], row):
                assert ze_utils.get_dropout_edit_string(str(sc), float(f)) == v, (sc, f)
        for sc, raises in zip(g["bad_schedules"], g["bad_raises"]):
            assert raises == 1
            with pytest.raises(Exception):
                ze_utils.get_dropout_edit_string(str(sc), 0.5)
    assert ze_utils.get_dropout_edit_string(None, 0.3) is None
    # SURVEY §8c example: '0,0@0.10,0.1@0.50,0' -> 0.05 at f = 0.3 and 0.75
    assert abs(ze_utils.get_dropout_edit_string("0,0@0.10,0.1@0.50,0", 0.3) - 0.05) < 1e-12
    assert abs(ze_utils.get_dropout_edit_string("0,0@0.10,0.1@0.50,0", 0.75) - 0.05) < 1e-12


def test_vectorised_chunk_table_equals_the_per_utterance_plan():
    """plan_chunk_table == plan_chunks utterance by utterance (order by length, stable), for the recipe settings, for
    chunk_size -1, for lengths around every boundary, and for the degenerate chunk_size < min_chunk_size."""
    rng = np.random.default_rng(0)
    lens = np.concatenate([rng.integers(0, 700, 400), [0, 24, 25, 26, 99, 100, 101, 199, 200, 201, 224, 225, 10000, 10024, 10025, 30001]])
    for min_chunk, chunk in ((25, 10000), (25, 100), (100, -1), (25, 200), (1, 7), (50, 20)):
        order, c_utt, c_start, c_len, seg = engine.plan_chunk_table(lens, min_chunk, chunk)
        plans = [engine.plan_chunks(int(t), min_chunk, chunk) for t in lens]
        ref_order = sorted((i for i, p in enumerate(plans) if p), key=lambda i: lens[i])
        assert order.tolist() == ref_order
        flat = [(u, s, n) for u in ref_order for s, n in plans[u]]
        assert list(zip(c_utt.tolist(), c_start.tolist(), c_len.tolist())) == flat
        assert seg.tolist() == [0] + list(np.cumsum([len(plans[u]) for u in ref_order]))


def test_egs_tar_round_trip(tmp_path):
    """examples_io twin: write_egs_tar -> TarFileDataLoader serves the minibatches in order as float16 with their labels,
    then (None, None); count matches; a label file of the wrong length is refused."""
    import examples_io
    rng = np.random.default_rng(0)
    mats = [rng.standard_normal((4, 20 + 3 * i, 5)).astype(np.float32) for i in range(5)]
    labels = rng.integers(0, 9, (5, 4)).astype(np.int32)
    tar = str(tmp_path / "egs.7.tar")
    examples_io.write_egs_tar(tar, mats, labels)
    dl = examples_io.TarFileDataLoader(tar, queue_size=2)
    assert dl.count == 5
    for m, l in zip(mats, labels):
        data, lab = dl.pop(timeout=10)
        assert data.dtype == np.float16 and np.array_equal(data, m.astype(np.float16)) and np.array_equal(lab, l)
    assert dl.pop() == (None, None)
    np.save(tar.replace(".tar", ".npy"), labels[:3])
    with pytest.raises(AssertionError):
        examples_io.TarFileDataLoader(tar)


def test_ranges_loader_serves_what_the_reference_loader_serves(tmp_path):
    """examples_io.RangesDataLoader == reference process_range_file + load_ranges_data + [shuffle] + DataLoader on the toy
    table of tests/golden/egs_ranges.npz: same minibatches, same order (plain and with --shuffle under seed 11)."""
    import examples_io
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "egs_ranges.npz"))
    keys = [str(k) for k in g["keys"]]
    ark, scp, ranges = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp"), str(tmp_path / "ranges.1")
    with kaldi_io.TableWriter(ark, scp) as tw:
        for i, k in enumerate(keys):
            kaldi_io.write_mat(tw, g["feat_%d" % i], key=k)
    open(ranges, "wt").write(str(g["ranges"]))
    count, B, F = int(g["count"]), int(g["minibatch_size"]), int(g["feat_dim"])
    for tag, shuffle, seed in (("plain", False, 0), ("shuffled", True, 11)):
        if seed:
            np.random.seed(seed)
        dl = examples_io.RangesDataLoader(ranges, scp, count, B, F, shuffle=shuffle)
        assert dl.count == count
        for i in range(count):
            d, l = dl.pop()
            assert d.dtype == np.float32 and np.array_equal(d, g["%s_data_%d" % (tag, i)]), (tag, i)
            assert np.array_equal(l, g["%s_labels_%d" % (tag, i)]), (tag, i)
        assert dl.pop() == (None, None)
    with pytest.raises(AssertionError):
        examples_io.RangesDataLoader(ranges, scp, count, B + 1, F)


def test_trainer_cli_accepts_the_command_line_train_dnn_builds(tmp_path):
    """The flag set train_dnn.py:269-301 puts on every trainer job (incl. the ones the model code never reads) parses, with
    the tar archive taking precedence over ranges/scp when both are given (train_dnn.py:264-267)."""
    import examples_io
    import train_dnn_one_iteration as cli
    from xvector_amd import synthetic, topology, weights as wio
    mdir = str(tmp_path / "model_3")
    topo = topology.get("ModelWithoutDropout")
    wio.save_model_dir(mdir, synthetic.reference_init(topo, 23, 8, seed=0), topo, "ModelWithoutDropout", 8, 23)
    tar = str(tmp_path / "egs.7.tar")
    examples_io.write_egs_tar(tar, [np.zeros((2, 30, 23), np.float32)], np.zeros((1, 2), np.int32))
    argv = ["--use-gpu=yes", "--verbose=0", "--print-interval=10", "--momentum=0.0", "--max-param-change=2.0",
            "--l2-regularize-factor=0.3333333333333333", "--random-seed=8", "--learning-rate=0.0015", "--scale=1.0",
            "--minibatch-count=1", "--feature-dim=23", "--dropout-proportion=0.05", "--tar-file=" + tar,
            "--ranges-file=" + str(tmp_path / "temp" / "ranges.7"), "--scp-file=" + str(tmp_path / "temp" / "feats.scp.7"), "--shuffle=True",
            "--minibatch-size=2", "--input-dir=" + mdir, "--output-dir=" + str(tmp_path / "model_4.1")]
    args = cli.get_args(argv)
    assert args.tar_file == tar and args.learning_rate == 0.0015 and args.dropout_proportion == 0.05 and args.random_seed == 8
    assert args.shuffle is True and args.minibatch_size == 2 and args.print_interval == 10
    with pytest.raises(Exception):
        cli.get_args([a for a in argv if not a.startswith("--tar-file")])          # no tar and the ranges file does not exist


def test_build_model_initialisers_per_class():
    """reference_init restates each class's initialisers: truncated_normal(0.1) / b = 0.1 / Xavier-uniform output
    (models.py:56-58,98-100); He-normal / He-uniform / Glorot for ModelL2LossWithoutDropoutReluHeInit (models.py:1158-1210);
    attention/{b,v} = 0.1, attention/w truncated_normal(0.1) (models.py:1040-1043)."""
    from xvector_amd import synthetic, topology
    w = synthetic.reference_init(topology.get("ModelWithoutDropout"), 23, 64, seed=3)
    k = w["frame_level_info_layer-1/w:0"]
    assert k.shape == (5, 512, 512) and abs(k.std() - 0.1 * 0.8796) < 2e-3 and np.abs(k).max() <= 0.2     # 2-sigma truncation
    assert np.all(w["frame_level_info_layer-1/b:0"] == np.float32(0.1)) and np.all(w["output/b:0"] == np.float32(0.1))
    assert np.abs(w["output/w:0"]).max() <= np.sqrt(6.0 / (512 + 64))
    he = synthetic.reference_init(topology.get("ModelL2LossWithoutDropoutReluHeInit"), 23, 64, seed=3)
    for name, fan_in in (("frame_level_info_layer-0", 5 * 23), ("frame_level_info_layer-2", 7 * 512), ("embed_layer-0", 3072)):
        k, b = he[name + "/w:0"], he[name + "/b:0"]
        sd = np.sqrt(2.0 / fan_in)
        assert abs(k.std() / (sd * 0.8796) - 1) < 0.03 and np.abs(k).max() <= 2 * sd * 1.0001
        assert np.abs(b).max() <= np.sqrt(6.0 / fan_in) and b.std() > 0.4 * np.sqrt(6.0 / fan_in)
    assert abs(he["output/w:0"].std() / (np.sqrt(2.0 / 576) * 0.8796) - 1) < 0.03
    att = synthetic.reference_init(topology.get("ModelL2LossWithoutDropoutLReluAttention"), 23, 64, seed=3)
    assert att["attention/w:0"].shape == (1536, 1536) and np.all(att["attention/v:0"] == np.float32(0.1))
    assert att["frame_level_info_layer-4/w:0"].shape == (1, 512, 3072) and att["embed_layer-0/w:0"].shape == (3072, 512)


def test_native_batch_packer_equals_numpy_pack():
    """xv_pack_rows_f32 (libxvector_host.so) writes exactly what BatchLayout.pack + row_valid write: chunk rows copied, every
    other row zeroed, padding columns untouched, on 1..5 threads; inconsistent layouts are refused."""
    from xvector_amd import engine
    lib = engine._host_lib()
    assert lib is not None
    rng = np.random.default_rng(0)
    for F, ld, align, lens in ((23, 24, 8, [25, 1, 300, 7, 64] * 9), (23, 24, 1, [5]), (16, 16, 1, [40, 3, 3, 900]), (23, 24, 8, [])):
        mats = [rng.standard_normal((t, F)).astype(np.float32) for t in lens]
        lay = engine.BatchLayout(lens, 3, align)
        want = np.full((lay.rows + 5, ld), 7.0, np.float32); want[:, F:] = 0
        want_rv = np.full(lay.rows + 5, 9, np.uint8)
        lay.pack(mats, want); lay.row_valid(want_rv)
        src = np.array([m.__array_interface__["data"][0] for m in mats], np.uint64)
        ln = np.array(lens, np.int32)
        for nt in (1, 2, 5):
            got = np.full((lay.rows + 5, ld), 7.0, np.float32); got[:, F:] = 0
            got_rv = np.full(lay.rows + 5, 9, np.uint8)
            rc = lib.xv_pack_rows_f32(src.ctypes.data, ln.ctypes.data, lay.row_start.ctypes.data, len(lens), F, got.ctypes.data, ld,
                                      lay.rows, got_rv.ctypes.data, nt)
            assert rc == 0 and np.array_equal(got, want) and np.array_equal(got_rv, want_rv), (F, ld, align, nt)
    bad = np.array([10, 5], np.int32)                                   # not ascending
    two = np.array([3, 3], np.int32)
    buf = np.zeros((32, 24), np.float32)
    assert lib.xv_pack_rows_f32(src.ctypes.data if len(src) else 0, two.ctypes.data, bad.ctypes.data, 2, 23, buf.ctypes.data, 24, 32, None, 1) == -1
    assert lib.xv_pack_rows_f32(0, two.ctypes.data, np.array([0, 30], np.int32).ctypes.data, 2, 23, buf.ctypes.data, 24, 32, None, 1) == -1   # overruns dst_rows


def test_cli_scp_sharding_helpers(tmp_path):
    """extract_embedding._scp_shard: contiguous line ranges that tile the table in order; the VAD scp follows the keys of the
    feature shard (missing VAD entries are simply absent -> the extractor warns and drops those keys)."""
    import extract_embedding as cli
    feats, vad = tmp_path / "feats.scp", tmp_path / "vad.scp"
    keys = ["utt%02d" % i for i in range(11)]
    feats.write_text("".join("%s feats.ark:%d\n" % (k, 100 * i) for i, k in enumerate(keys)) + "\n")
    vad.write_text("".join("%s vad.ark:%d\n" % (k, 7 * i) for i, k in reversed(list(enumerate(keys))) if k != "utt04"))
    assert cli._is_scp_table("scp:x.scp") and cli._is_scp_table("scp,s,cs: x.scp") and not cli._is_scp_table("ark:x.ark")
    assert not cli._is_scp_table("scp:cat x.scp |")
    seen = []
    for world in (1, 3, 4):
        got = []
        for r in range(world):
            f, v, all_keys = cli._scp_shard("scp:%s" % feats, r, world, "scp:%s" % vad)
            fk = [ln.split()[0] for ln in f]                             # the shard's lines (kaldi_io.MatScp takes them as they are)
            assert all_keys[r] == fk and sum((list(k) for k in all_keys), []) == keys   # every rank knows every shard's keys: no key exchange
            vk = [ln.split()[0] for ln in v]
            assert vk == [k for k in fk if k != "utt04"]                  # same order as the feature shard
            got += fk
        assert got == keys
        seen.append(got)
    f, v, all_keys = cli._scp_shard("scp:%s" % feats, 0, 2)
    assert v is None and len(f) == 5 and [len(k) for k in all_keys] == [5, 6]


def test_batches_come_in_whole_rounds_of_workgroups():
    """Extractor._batch_bounds: a window larger than one batch is cut into the fewest batches, sized in whole rounds (32768 rows)
    of resident workgroups -- the rounds paid equal the rounds the window needs -- every batch within the row budget, every
    chunk in exactly one batch, in order; small budgets (tests, tiny batches) keep the plain even / greedy split."""
    from xvector_amd import engine
    ex = engine.Extractor.__new__(engine.Extractor)
    ex.max_batch_rows, ex.max_batch_chunks = 262144, 8192
    rng = np.random.default_rng(0)
    for n_utt in (100, 900, 2300, 5000, 30000):
        lens = np.sort(rng.integers(200, 401, size=n_utt))
        cum = np.zeros(n_utt + 1, np.int64)
        np.cumsum(engine.slot_rows(lens, 3, 8), out=cum[1:])
        bounds = ex._batch_bounds(cum, 8)
        assert bounds[0][0] == 0 and bounds[-1][1] == n_utt and all(a[1] == b[0] for a, b in zip(bounds[:-1], bounds[1:]))
        rows = [r for _, _, r in bounds]
        assert all(r == 8 + int(cum[b1] - cum[b0]) for (b0, b1, r) in bounds) and max(rows) <= 262144
        paid = sum(-(-r // ex.ROUND_ROWS) for r in rows)
        need = -(-(int(cum[-1]) + 8 * len(rows)) // ex.ROUND_ROWS)
        assert paid <= need, (n_utt, rows)
    # a budget below two rounds: the old behaviour (even split, one chunk may exceed the budget)
    ex.max_batch_rows = 700
    cum = np.concatenate([[0], np.cumsum([304, 304, 808, 96, 96])])
    bounds = ex._batch_bounds(cum, 8)
    assert [b[:2] for b in bounds] == [(0, 2), (2, 3), (3, 5)]


def test_raw_row_plan_matches_the_array_formulation():
    """xv_raw_row_plan (libxvector_host.so): the destination row of every raw frame of a batch -- voiced frames counted per
    utterance, cut into the plan's chunks, chunks of other batches and dropped tails marked -1 -- against the same rule written
    with whole-array operations (what Extractor.submit_raw did before), on random windows with and without a VAD."""
    import ctypes
    from xvector_amd import engine
    lib = engine._host_lib()
    if lib is None:
        pytest.skip("host library not built")
    rng = np.random.default_rng(8)
    for trial in range(60):
        nU = int(rng.integers(1, 40))
        T = rng.integers(1, 300, nU).astype(np.int64)
        with_vad = trial % 3 != 0
        vstart = np.zeros(nU, np.int64)
        np.cumsum(T[:-1], out=vstart[1:])
        voiced = (rng.random(int(T.sum())) < rng.choice([0.2, 0.8, 1.0])) if with_vad else None
        V = np.add.reduceat(voiced, vstart).astype(np.int64) if with_vad else T.copy()
        size = np.maximum(1, rng.integers(20, 120, nU)).astype(np.int64)
        kept = np.minimum(rng.integers(0, 5, nU), -(-V // size)).astype(np.int64)          # chunks of the utterance that exist
        seg = np.zeros(nU, np.int64)
        np.cumsum(kept[:-1], out=seg[1:])
        nch = int(kept.sum())
        b0 = int(rng.integers(0, max(1, nch // 2 + 1)))
        b1 = int(rng.integers(b0, nch + 1))
        row_start = (np.arange(max(b1 - b0, 1)) * 137 + 5).astype(np.int32)
        # whole-array formulation
        p = np.repeat(np.arange(nU), T)
        if voiced is None:
            vm = np.ones(int(T.sum()), bool)
            vidx = np.arange(int(T.sum()), dtype=np.int64) - np.repeat(vstart, T)
        else:
            vm = voiced
            cv = np.cumsum(vm, dtype=np.int64)
            before = cv[vstart + T - 1] - V
            vidx = cv - 1 - np.repeat(before, T)
        k = vidx // size[p]
        cid = seg[p] + k
        ok = vm & (k < kept[p]) & (cid >= b0) & (cid < b1)
        want = np.where(ok, row_start[np.clip(cid - b0, 0, len(row_start) - 1)] + (vidx - k * size[p]), -1).astype(np.int32)
        got = np.full(int(T.sum()) + 3, 12345, np.int32)
        fl = np.ascontiguousarray(voiced) if voiced is not None else None
        rc = lib.xv_raw_row_plan(nU, T.ctypes.data, fl.ctypes.data if fl is not None else None, vstart.ctypes.data, size.ctypes.data,
                                 kept.ctypes.data, seg.ctypes.data, b0, b1, row_start.ctypes.data, got.ctypes.data, int(T.sum()))
        assert rc == 0 and np.array_equal(got[:-3], want) and np.all(got[-3:] == 12345), trial
        assert lib.xv_raw_row_plan(nU, T.ctypes.data, None, vstart.ctypes.data, size.ctypes.data, kept.ctypes.data, seg.ctypes.data, b0, b1,
                                   row_start.ctypes.data, got.ctypes.data, int(T.sum()) - 1) == -1


def test_select_voiced_window_rules_match_the_per_utterance_oracle(oracle_mod):
    """frontend.select_voiced (the utterance-level rules of select-voiced-frames for a whole window, on concatenated arrays) against
    the oracle's per-utterance restatement: a VAD of another length or without a voiced frame drops the utterance, no VAD means every
    frame, an utterance with neither frames nor a VAD stays an empty matrix; flags of the surviving utterances in order."""
    from xvector_amd import frontend
    rng = np.random.default_rng(4)
    for trial in range(40):
        n = int(rng.integers(1, 60))
        mats, vads = [], []
        for i in range(n):
            T = int(rng.choice([0, 1, 5, 40, 300], p=[.05, .1, .15, .4, .3]))
            mats.append(rng.standard_normal((T, 5)).astype(np.float32))
            r = rng.random()
            if r < 0.15:
                vads.append(None)
            elif r < 0.25:
                vads.append(np.zeros(T, np.float32))                                   # nothing voiced
            elif r < 0.35:
                vads.append((rng.random(max(0, T + int(rng.choice([-1, 1, 7])))) < 0.7).astype(np.float32))       # another length
            else:
                vads.append((rng.random(T) < rng.choice([0.3, 0.9, 1.0])).astype(np.float32).reshape(-1, 1) if trial % 2 else
                            (rng.random(T) < 0.8).astype(np.float32))
        if trial % 5 == 0:
            vads_in = None
        else:
            vads_in = vads
        T, cand, voiced, empty, dropped = frontend.select_voiced(mats, vads_in)
        assert T.tolist() == [m.shape[0] for m in mats]
        want_cand, want_flags, want_empty, want_dropped = [], [], [], 0
        for i, m in enumerate(mats):
            v = None if vads_in is None else vads_in[i]
            if v is None:
                if m.shape[0] > 0:
                    want_cand.append(i)
                    want_flags.append(np.ones(m.shape[0], bool))
                else:
                    want_empty.append(i)
                continue
            v1 = np.asarray(v).reshape(-1)
            sel = oracle_mod.select_voiced(m, v1) if m.shape[0] > 0 else None
            if sel is None:
                want_dropped += 1
            else:
                want_cand.append(i)
                want_flags.append(v1 != 0)
                assert np.array_equal(sel, m[v1 != 0])
        assert cand.tolist() == want_cand and sorted(empty) == want_empty and dropped == want_dropped, trial
        if vads_in is None:
            assert voiced is None
        else:
            assert np.array_equal(voiced, np.concatenate(want_flags) if want_flags else np.zeros(0, bool)), trial


def test_vad_runs_select_like_the_list_of_vectors():
    """frontend.VadRuns -- the VAD vectors of a window as a few (values, offsets) runs, what the reader hands over while the VAD table's
    keys arrive in the order of the features -- gives select_voiced's answers for the list of per-utterance vectors it stands for, and
    reads back as that list."""
    from xvector_amd import frontend
    rng = np.random.default_rng(9)
    for trial in range(40):
        n = int(rng.integers(1, 80))
        mats, vads = [], []
        for i in range(n):
            T = int(rng.choice([0, 1, 5, 40, 300], p=[.05, .1, .15, .4, .3]))
            mats.append(rng.standard_normal((T, 3)).astype(np.float32))
            r = rng.random()
            if r < 0.1 and trial % 3:
                vads.append(np.zeros(T, np.float32))                                   # nothing voiced
            elif r < 0.2 and trial % 3:
                vads.append((rng.random(max(0, T + int(rng.choice([-1, 1, 7])))) < 0.7).astype(np.float32))       # another length
            else:
                vads.append((rng.random(T) < rng.choice([0.3, 0.9, 1.0])).astype(np.float32))
        runs = frontend.VadRuns()
        i = 0
        while i < n:                                       # runs of random lengths, now and then a single vector
            m = int(rng.integers(1, 12))
            part = vads[i:i + m]
            if m == 1 or rng.random() < 0.2:
                for v in part:
                    runs.add_one(v)
            else:
                offs = np.concatenate([[0], np.cumsum([len(v) for v in part])])
                runs.add_run(np.concatenate(part) if part else np.zeros(0, np.float32), offs)
            i += m
        assert len(runs) == n and all(np.array_equal(a, b) for a, b in zip(runs, vads)) and np.array_equal(runs[n - 1], vads[n - 1])
        got, want = frontend.select_voiced(mats, runs), frontend.select_voiced(mats, vads)
        assert got[0].tolist() == want[0].tolist() and got[1].tolist() == want[1].tolist(), trial
        assert np.array_equal(got[2], want[2]) and list(got[3]) == list(want[3]) and got[4] == want[4], trial
