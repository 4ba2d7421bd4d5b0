"""GPU parity of the whole path: DeviceModel / Extractor / Model.make_embedding vs golden fixtures and vs the oracle run live on
small cases.  The golden x-vectors (tests/golden/forward_refgraph.npz) are what the REFERENCE'S OWN GRAPHS return: every build_model
of local/tf/models.py executed under tests/golden/numpy_tf1.py (NumPy float64 evaluation of the TF ops), loaded by the reference's
load_model, fed as make_embedding feeds it (tests/golden/make_golden.py).

Bar (BASELINE.json north_star): embeddings within 1e-4 relative L2 of the CPU reference on identical
inputs.  The tests assert the much tighter 1e-5 that exact-fp32 MFMA accumulation delivers.
"""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-5, "fp32tc": 1e-5, "bf16x3": 5e-5, "f16bf8": 6e-5}      # asserted; the north-star bar is 1e-4 for every path
PRECISIONS = ["fp32", "bf16x3", "f16bf8"]


@pytest.fixture(scope="module")
def env(oracle_mod):
    import torch
    from xvector_amd import engine, hiplib, synthetic, topology
    hiplib.require_gpu()
    return dict(torch=torch, hiplib=hiplib, engine=engine, oracle=oracle_mod, synthetic=synthetic, topology=topology)


@pytest.mark.parametrize("tname,cls", [("default", "ModelWithoutDropout"), ("dilated", "ModelWithoutDropoutTdnn"),
                                       ("prelu", "ModelWithoutDropoutPRelu"), ("lrelu", "ModelL2LossWithoutDropoutLRelu"),
                                       ("attention", "ModelL2LossWithoutDropoutLReluAttention"), ("dropout", "Model"),
                                       ("l2prelu", "ModelL2LossWithoutDropoutPRelu"), ("heinit", "ModelL2LossWithoutDropoutReluHeInit")])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_matches_golden(env, golden, tname, cls, precision):
    """All 8 classes of local/tf/models.py: the HIP path vs the vectors the reference's own graph returned."""
    g = golden("forward_refgraph.npz")
    seed = int(g["seed"])
    topo = env["topology"].get(cls)
    w = env["synthetic"].trained_like(topo, 23, seed=seed)
    rng = np.random.default_rng(seed + 1)
    Ts = [25, 200, 400, 1000] if tname in ("default", "dilated") else [25, 200, 1000] if tname == "attention" else [25, 200]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in Ts]
    for emb_idx in (0, 1):
        model = env["engine"].DeviceModel(w, topo, "cuda:0", embedding_index=emb_idx, precision=precision)
        ex = env["engine"].Extractor(model, 1, -1)
        vecs = ex.extract(mats)
        for T, v in zip(Ts, vecs):
            ref = g["%s_T%d_e%d" % (tname, T, emb_idx)]
            # single chunk: the reference's (T*e)/T float32 round trip is part of the contract
            ref32 = env["oracle"].chunk_average(ref.astype(np.float32)[None, :], [T], np.float32)
            assert env["oracle"].rel_l2(v, ref32) < TOL[precision], (tname, T, emb_idx)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_layer_intermediates_T25(env, golden, precision):
    """Per-layer tensors for a T=25 utterance: localises edge-padding bugs (SURVEY §7.3 hard part 1)."""
    torch = env["torch"]
    g = golden("forward_refgraph.npz")
    seed = int(g["seed"])
    topo = env["topology"].get("ModelWithoutDropout")
    w = env["synthetic"].trained_like(topo, 23, seed=seed)
    rng = np.random.default_rng(seed + 1)
    x = (rng.standard_normal((25, 23)) * 3.0).astype(np.float32)
    model = env["engine"].DeviceModel(w, topo, "cuda:0", precision=precision)
    # surround the utterance with other chunks so that wrong neighbours WOULD leak in
    other = (rng.standard_normal((40, 23)) * 3.0).astype(np.float32)
    layout = env["engine"].BatchLayout([40, 25, 40], model.gap)
    host = np.zeros((layout.rows, model.in_dim), np.float32)
    layout.pack([other, x, other], host)
    xd = torch.from_numpy(host).cuda()
    rv = torch.from_numpy(layout.row_valid()).cuda()
    outs = model.intermediates_packed(xd, rv)
    s = layout.row_start[1]
    for li, y in enumerate(outs):
        got = y.cpu().numpy()[s:s + 25, ::16]
        assert env["oracle"].rel_l2(got, g["default_T25_layer%d_sub" % li]) < TOL[precision], li


@pytest.mark.parametrize("precision", PRECISIONS)
def test_extractor_ragged_batch_vs_oracle(env, default_weights, precision):
    """Config-3 style: lengths from 25 to several thousand, chunking on, small batch budget so that
    several batches and the pooling split path are exercised; checked against the fp32/fp64 oracle."""
    topo, w = default_weights
    oracle = env["oracle"]
    rng = np.random.default_rng(77)
    lens = [25, 26, 31, 100, 257, 999, 1024, 2300, 24, 0, 613]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in lens]
    model = env["engine"].DeviceModel(w, topo, "cuda:0", precision=precision)
    ex = env["engine"].Extractor(model, 25, 1000, max_batch_rows=1500)
    vecs = ex.extract(mats)
    assert ex.stats["batches"] > 2
    for T, m, v in zip(lens, mats, vecs):
        ref = oracle.embed_utterance(m, w, topo, 25, 1000, np.float64)
        if T < 25:
            assert v is None and ref is None
            continue
        assert oracle.rel_l2(v, ref) < TOL[precision], T


def test_fused_and_standalone_pooling_agree(env, default_weights):
    """bf16x3: the pooling reduction in the last GEMM's epilogue (default) vs storing the layer and running the standalone
    pooling kernel -- same GEMM, two reduction orders: agree to fp32 rounding, both within tolerance of the oracle."""
    topo, w = default_weights
    oracle = env["oracle"]
    rng = np.random.default_rng(5)
    lens = [25, 200, 333, 400, 1203, 64]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in lens]
    res = {}
    for fused in (True, False):
        model = env["engine"].DeviceModel(w, topo, "cuda:0", precision="bf16x3", fused_pool=fused)
        assert model.align == (8 if fused else 1)
        res[fused] = env["engine"].Extractor(model, 25, 1000, max_batch_rows=1500).extract(mats)
    for m, a, b in zip(mats, res[True], res[False]):
        assert oracle.rel_l2(a, b) < 2e-6
        assert oracle.rel_l2(a, oracle.embed_utterance(m, w, topo, 25, 1000, np.float64)) < TOL["bf16x3"]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_batch_composition_does_not_change_bits(env, default_weights, precision):
    """Size-independent property at full width: an utterance's x-vector is bit-identical whether it is
    extracted alone or inside any batch, in any order (the reference is batch-1 by construction)."""
    topo, w = default_weights
    rng = np.random.default_rng(123)
    lens = rng.integers(200, 401, size=40)
    mats = [(rng.standard_normal((int(T), 23)) * 3.0).astype(np.float32) for T in lens]
    model = env["engine"].DeviceModel(w, topo, "cuda:0", precision=precision)
    ex = env["engine"].Extractor(model, 25, 10000)
    together = ex.extract(mats)
    perm = rng.permutation(len(mats))
    shuffled = ex.extract([mats[i] for i in perm])
    for j, i in enumerate(perm):
        assert np.array_equal(together[i], shuffled[j])
    for i in (0, 7, 39):
        assert np.array_equal(ex.extract([mats[i]])[0], together[i])


@pytest.mark.parametrize("precision", PRECISIONS + ["fp32tc"])
def test_make_embedding_end_to_end_control_flow(env, golden, tmp_path, precision, monkeypatch):
    """Model.make_embedding (twin of models.py:356-432) ark->ark on the control-flow fixture: the
    output stream has the reference's keys, order and framing; vectors within tolerance of the bytes
    the REFERENCE driver produced (under a stub TF session evaluating the fp64 oracle)."""
    import logging
    import kaldi_io
    import models
    from fixture_inputs import CONTROL_FEAT, CONTROL_SEED, control_inputs
    monkeypatch.setenv("XVECTOR_PRECISION", precision)
    g = golden("make_embedding.npz")
    topo = env["synthetic"].SMALL_TOPOLOGY
    w = env["synthetic"].trained_like(topo, CONTROL_FEAT, num_classes=8, seed=CONTROL_SEED)
    mdir = str(tmp_path / "model_small")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=8, feat_dim=CONTROL_FEAT),
                            mdir, None)
    bio = io.BytesIO()
    for k, m in control_inputs():
        kaldi_io.write_mat(bio, m, key=k)
    in_ark = bio.getvalue()
    log = logging.getLogger("test_make_embedding")
    for si, (min_chunk, chunk) in enumerate(g["settings"]):
        out = io.BytesIO()
        models.Model().make_embedding(io.BytesIO(in_ark), out, mdir, int(min_chunk), int(chunk), False, log)
        got = list(kaldi_io.read_vec_flt_ark(io.BytesIO(out.getvalue())))
        ref = list(kaldi_io.read_vec_flt_ark(io.BytesIO(g["out_ark_%d" % si].tobytes())))
        assert [k for k, _ in got] == [k for k, _ in ref]
        assert len(out.getvalue()) == len(g["out_ark_%d" % si])
        for (k, a), (_, b) in zip(got, ref):
            assert a.dtype == np.float32 and a.shape == b.shape
            assert env["oracle"].rel_l2(a, b) < TOL[precision], (k, min_chunk, chunk)


@pytest.mark.parametrize("precision", PRECISIONS + ["fp32tc"])
def test_make_embedding_matches_the_reference_graph_stream(env, golden, tmp_path, precision, monkeypatch):
    """ark bytes in -> ark bytes out, the reference on both ends: forward_refgraph.npz embed_out_ark_* is the stream the reference's
    make_embedding (models.py:356-432) wrote while driving the reference's own ModelWithoutDropout graph (numpy_tf1 evaluation) on a
    model directory its own build_model created.  The twin's make_embedding on the HIP path: same keys, same order, same framing, same
    length, vectors within the asserted tolerance -- the whole chain with no statement of mine between the reference and the check."""
    import logging
    import kaldi_io
    import models
    monkeypatch.setenv("XVECTOR_PRECISION", precision)
    g = golden("forward_refgraph.npz")
    seed = int(g["seed"])
    topo = env["topology"].get("ModelWithoutDropout")
    w = env["synthetic"].trained_like(topo, 23, seed=seed)
    mdir = str(tmp_path / "model_default")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=64, feat_dim=23), mdir, None)
    rng = np.random.default_rng(seed + 2)
    bio = io.BytesIO()
    for i, T in enumerate(g["embed_lengths"]):
        kaldi_io.write_mat(bio, (rng.standard_normal((int(T), 23)) * 3.0).astype(np.float32), key="rg%02d-T%d" % (i, T))
    log = logging.getLogger("test_refgraph_stream")
    for si, (min_chunk, chunk) in enumerate(g["embed_settings"]):
        out = io.BytesIO()
        models.ModelWithoutDropout().make_embedding(io.BytesIO(bio.getvalue()), out, mdir, int(min_chunk), int(chunk), False, log)
        want = g["embed_out_ark_%d" % si].tobytes()
        got = list(kaldi_io.read_vec_flt_ark(io.BytesIO(out.getvalue())))
        ref = list(kaldi_io.read_vec_flt_ark(io.BytesIO(want)))
        assert [k for k, _ in got] == [k for k, _ in ref] and len(out.getvalue()) == len(want)
        for (k, a), (_, b) in zip(got, ref):
            assert a.dtype == b.dtype == np.float32
            assert env["oracle"].rel_l2(a, b) < TOL[precision], (k, min_chunk, chunk)


def test_in_memory_stream_and_arena_reader_give_the_same_bytes(env, tmp_path, monkeypatch):
    """An io.BytesIO is scanned where it lies (kaldi_io.map_stream), every other stream is read into the arenas, and
    XVECTOR_MAP_INPUT=0 sends a BytesIO down the arena path too: the three give byte-identical output arks -- also when the
    stream holds a record the native scanner does not take (one double-precision matrix in the middle) and utterances that
    are rejected (T < min_chunk)."""
    import logging
    import kaldi_io
    import models
    topo = env["synthetic"].SMALL_TOPOLOGY
    w = env["synthetic"].trained_like(topo, 5, num_classes=8, seed=3)
    mdir = str(tmp_path / "m")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=8, feat_dim=5), mdir, None)
    rng = np.random.default_rng(9)
    bio = io.BytesIO()
    for i in range(400):
        T = int(rng.integers(5, 300))
        m = (rng.standard_normal((T, 5)) * 2).astype(np.float32)
        kaldi_io.write_mat(bio, m.astype(np.float64) if i == 123 else m, key="u%03d" % i)
    data = bio.getvalue()
    log = logging.getLogger("test_streams")

    class Stream(io.BytesIO):                     # not exactly a BytesIO: goes through readinto into the arenas
        pass

    outs = []
    for make, env_value in ((io.BytesIO, "1"), (Stream, "1"), (io.BytesIO, "0")):
        monkeypatch.setattr(models.Model, "map_input", env_value != "0")
        src, out = make(data), io.BytesIO()
        models.Model().make_embedding(src, out, mdir, 25, 10000, False, log)
        outs.append(out.getvalue())
        src.close()
    assert len(outs[0]) > 0 and outs[0] == outs[1] == outs[2]
    keys = [k for k, _ in kaldi_io.read_vec_flt_ark(io.BytesIO(outs[0]))]
    assert "u123" in keys and len(keys) < 400      # the double record came through; the short utterances were rejected


def test_out_of_range_windows_in_an_arena_stream_are_repeated_from_live_bytes(env, default_weights, tmp_path, monkeypatch):
    """make_embedding over a stream that is read into recycled arenas (not a BytesIO), with windows whose activations leave
    the fp16 range: ``finish`` repeats such a window on the bf16x3 twin one loop iteration later, packing it again from the raw
    addresses -- the arena must still hold that window (it may only go back to the pool after the window is collected; with
    the reader ahead of the GPU it was being refilled by then and the repeat silently packed another window's bytes)."""
    import logging
    import kaldi_io
    import models
    topo, w = default_weights
    mdir = str(tmp_path / "m")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=8, feat_dim=23), mdir, None)
    rng = np.random.default_rng(77)
    n = 1400
    wild = set(int(i) for i in rng.choice(n, 12, replace=False)) | {3, 4, 700}
    bio = io.BytesIO()
    for i in range(n):
        m = (rng.standard_normal((int(rng.integers(60, 140)), 23)) * 3.0).astype(np.float32)
        kaldi_io.write_mat(bio, m * np.float32(1e6) if i in wild else m, key="u%04d" % i)
    data = bio.getvalue()

    class Stream(io.BytesIO):                     # not exactly a BytesIO: read into the arenas
        pass
    log = logging.getLogger("test_arena_redo")
    monkeypatch.setattr(models.Model, "arena_bytes", 256 << 10)          # ~28 utterances per window, ~50 windows, 5 arenas
    monkeypatch.setattr(models.Model, "first_arena_bytes", 256 << 10)
    outs = {}
    for precision in ("f16bf8", "bf16x3"):
        monkeypatch.setenv("XVECTOR_PRECISION", precision)
        out = io.BytesIO()
        model = models.Model()
        model.make_embedding(Stream(data), out, mdir, 25, 10000, False, log)
        outs[precision] = dict(kaldi_io.read_vec_flt_ark(io.BytesIO(out.getvalue())))
        assert len(outs[precision]) == n
    a, b = outs["f16bf8"], outs["bf16x3"]
    worst = max(env["oracle"].rel_l2(a[k], b[k]) for k in a)
    assert worst < 1e-4, "f16bf8 stream vs bf16x3 stream: %.3e (a repeated window packed stale bytes?)" % worst
    for i in wild:                                # a repeated window IS the bf16x3 result of its own utterances ...
        k = "u%04d" % i
        assert np.isfinite(a[k]).all() and env["oracle"].rel_l2(a[k], b[k]) < 1e-5
    assert any(not np.array_equal(a[k], b[k]) for k in a)     # ... and the other windows did run in f16bf8


def test_config3_lengths_25_to_10000_full_topology(env, default_weights):
    """BASELINE config 3: T from 25 to 10000 (and beyond, to exercise chunking at chunk_size=10000), length-bucketed
    batches, default topology; parity on a subsample that includes T=25 and T=10000 against the fp32 C oracle
    (OpenMP; the fp64 one would take minutes here) plus the fp64 oracle for the short ones."""
    topo, w = default_weights
    oracle = env["oracle"]
    rng = np.random.default_rng(2025)
    lens = [25, 10000, 4999, 777, 10025, 63, 2048]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in lens]
    for precision in PRECISIONS:
        model = env["engine"].DeviceModel(w, topo, "cuda:0", precision=precision)
        ex = env["engine"].Extractor(model, 25, 10000, max_batch_rows=16384)
        vecs = ex.extract(mats)
        assert ex.stats["batches"] >= 3
        for T, m, v in zip(lens, mats, vecs):
            ref = oracle.embed_utterance(m, w, topo, 25, 10000, np.float64 if T <= 777 else np.float32)
            assert oracle.rel_l2(v, ref) < (TOL[precision] if T <= 777 else 5e-5), (precision, T)


def test_extract_embedding_cli_ark_scp(env, tmp_path):
    """The CLI twin end to end on files: ark in, `ark,scp:` out (tmp files renamed, scp patched), second call is
    a no-op because both outputs exist (extract_embedding.py:126-128 of the reference)."""
    import kaldi_io
    import models
    import extract_embedding as ee
    topo = env["synthetic"].SMALL_TOPOLOGY
    w = env["synthetic"].trained_like(topo, 5, num_classes=8, seed=11)
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=8, feat_dim=5), mdir, None)
    rng = np.random.default_rng(0)
    utts = [("u%02d" % i, (rng.standard_normal((T, 5)) * 3).astype(np.float32)) for i, T in enumerate([30, 10, 250, 99])]
    feats = tmp_path / "feats.ark"
    with open(feats, "wb") as f:
        for k, m in utts:
            kaldi_io.write_mat(f, m, key=k)
    ark, scp = tmp_path / "xv.ark", tmp_path / "xv.scp"
    argv = ["--use-gpu", "no", "--min-chunk-size", "25", "--chunk-size", "100", "--feature-rspecifier", "ark:%s" % feats,
            "--vector-wspecifier", "ark,scp:%s,%s" % (ark, scp), "--model-dir", mdir]
    ee.main(argv)
    assert ark.exists() and scp.exists() and not (tmp_path / "xv.ark.tmp.ark").exists()
    got = dict(kaldi_io.read_vec_flt_scp(str(scp)))
    assert list(got) == ["u00", "u02", "u03"]                       # u01 (T=10 < 25) is rejected, order kept
    assert all(str(ark) in line and ".tmp" not in line for line in open(scp).read().splitlines())
    for k, m in utts:
        if k in got:
            ref = env["oracle"].embed_utterance(m, w, topo, 25, 100, np.float64)
            assert env["oracle"].rel_l2(got[k], ref) < TOL["bf16x3"]
    before = ark.stat().st_mtime_ns
    ee.main(argv)                                                    # outputs exist -> returns immediately
    assert ark.stat().st_mtime_ns == before


def test_single_rank_process_group_paths(env, tmp_path):
    """Exercise the torch.distributed (RCCL) code path that the 8-GPU runs use, with the one GPU this box has:
    bench.py under torchrun with a forced 1-rank process group, and Model.make_embedding's sharded branch."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    envv = dict(os.environ, XV_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29100 + os.getpid() % 500), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1",
           "--warmup", "0", "--utts", "600", "--cpu-budget", "0", "--no-extra-legs", "--no-fp32-leg", "--e2e-utts", "0"]
    # (the process group is what is under test: the line's other legs -- 50 k utterances ark to ark, a CLI job, a training run --
    # have their own tests in test_gpu_bench_contract.py; with them this test was 315 of the suite's 590 seconds)
    out = subprocess.run(cmd, env=envv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["dist_initialized"] is True


def test_make_embedding_from_tf_checkpoint_dir(env, tmp_path):
    """A model directory in the reference's own on-disk format (TF1 Saver bundle, class name in model_name.txt)
    produces exactly the bytes the native weight container produces."""
    import logging
    import kaldi_io
    import models
    from test_tf_checkpoint import _tf_style_arrays, write_bundle
    topo = env["topology"].get("ModelWithoutDropout")
    w = env["synthetic"].trained_like(topo, 23, num_classes=12, seed=8)
    native = str(tmp_path / "native")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=12, feat_dim=23),
                            native, None)
    nnet = tmp_path / "nnet"
    tfdir = nnet / "model_final"
    tfdir.mkdir(parents=True)
    (nnet / "model_name.txt").write_text("ModelWithoutDropout")
    (tfdir / "model.meta").write_bytes(b"\x0a\x05graph")
    (tfdir / "done").write_text("done")
    write_bundle(str(tfdir / "model"), _tf_style_arrays(w), block_entries=9)
    rng = np.random.default_rng(5)
    bio = io.BytesIO()
    for i, T in enumerate([40, 300, 25]):
        kaldi_io.write_mat(bio, (rng.standard_normal((T, 23)) * 3).astype(np.float32), key="k%d" % i)
    log = logging.getLogger("tfdir")
    outs = []
    for d in (native, str(tfdir)):
        out = io.BytesIO()
        models.Model().make_embedding(io.BytesIO(bio.getvalue()), out, d, 25, 10000, False, log)
        outs.append(out.getvalue())
    assert outs[0] == outs[1] and len(outs[0]) == 3 * (3 + 2 + 3 + 5 + 512 * 4)


def test_make_embedding_empty_and_all_rejected_streams(env, tmp_path):
    """Edge cases of the driver loop (models.py:373-387): an empty feature stream and a stream in which every utterance
    is rejected produce an empty output stream and the summary log line, not an error; Extractor.extract([]) == []."""
    import logging
    import kaldi_io
    import models
    topo = env["synthetic"].SMALL_TOPOLOGY
    w = env["synthetic"].trained_like(topo, 5, num_classes=8, seed=3)
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=8, feat_dim=5), mdir, None)
    log = logging.getLogger("edge")
    out = io.BytesIO()
    models.Model().make_embedding(io.BytesIO(b""), out, mdir, 25, 100, False, log)
    assert out.getvalue() == b""
    bio = io.BytesIO()
    for i, T in enumerate([0, 3, 24]):
        kaldi_io.write_mat(bio, np.zeros((T, 5), np.float32), key="short%d" % i)
    out = io.BytesIO()
    models.Model().make_embedding(io.BytesIO(bio.getvalue()), out, mdir, 25, 100, False, log)
    assert out.getvalue() == b""
    model = env["engine"].DeviceModel(w, topo, "cuda:0")
    ex = env["engine"].Extractor(model, 25, 100)
    assert ex.extract([]) == [] and ex.extract([np.zeros((3, 5), np.float32)]) == [None]
    # one enormous chunk (chunk_size -1) larger than the batch budget still goes through as a batch of its own
    big = (np.random.default_rng(0).standard_normal((5000, 5)) * 2).astype(np.float32)
    v = env["engine"].Extractor(model, 25, -1, max_batch_rows=1024).extract([big])[0]
    assert env["oracle"].rel_l2(v, env["oracle"].embed_utterance(big, w, topo, 25, -1, np.float64)) < TOL["bf16x3"]


def test_config1_100_utterances_t200_through_the_twin(env, default_weights, tmp_path):
    """BASELINE configs[0] on the GPU path: the same 100 x [200, 23] utterances (seed 1234) ark -> Model.make_embedding ->
    ark, every x-vector against the fp64 oracle for a sample and against the fp32 C oracle for all."""
    import logging
    import kaldi_io
    import models
    topo, w = default_weights
    oracle = env["oracle"]
    rng = np.random.default_rng(1234)
    utts = [("utt%06d" % i, (rng.standard_normal((200, 23)) * 3.0).astype(np.float32)) for i in range(100)]
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=64, feat_dim=23), mdir, None)
    bio = io.BytesIO()
    for k, m in utts:
        kaldi_io.write_mat(bio, m, key=k)
    out = io.BytesIO()
    models.Model().make_embedding(io.BytesIO(bio.getvalue()), out, mdir, 25, 10000, False, logging.getLogger("cfg1"))
    got = list(kaldi_io.read_vec_flt_ark(io.BytesIO(out.getvalue())))
    assert [k for k, _ in got] == [k for k, _ in utts]
    worst32 = max(oracle.rel_l2(v, oracle.embed_utterance(m, w, topo, 25, 10000, np.float32)) for (_, v), (_, m) in zip(got, utts))
    worst64 = max(oracle.rel_l2(got[i][1], oracle.embed_utterance(utts[i][1], w, topo, 25, 10000, np.float64)) for i in (0, 37, 99))
    assert worst32 < TOL["bf16x3"] and worst64 < TOL["bf16x3"]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_input_scale_and_degenerate_utterances(env, default_weights, precision):
    """The split-precision path inherits fp32's exponent range (bf16 hi + bf16 lo): features 1000x larger or smaller than
    MFCCs, an all-zero utterance (every channel constant over time -> std = sqrt(1e-5) exactly) and a constant non-zero
    utterance stay within tolerance of the fp64 oracle."""
    topo, w = default_weights
    oracle = env["oracle"]
    rng = np.random.default_rng(17)
    base = (rng.standard_normal((180, 23)) * 3.0).astype(np.float32)
    mats = [base * np.float32(1e3), base * np.float32(1e-3), np.zeros((64, 23), np.float32), np.full((90, 23), 2.5, np.float32), base]
    model = env["engine"].DeviceModel(w, topo, "cuda:0", precision=precision)
    vecs = env["engine"].Extractor(model, 25, 10000).extract(mats)
    for m, v in zip(mats, vecs):
        ref = oracle.embed_utterance(m, w, topo, 25, 10000, np.float64)
        assert np.isfinite(v).all() and oracle.rel_l2(v, ref) < TOL[precision]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_attention_pooling_ragged_batch_and_chunking(env, precision):
    """ModelL2LossWithoutDropoutLReluAttention (models.py:985-1114) through the extractor: ragged batch with chunked long
    utterances (softmax runs over the frames of ONE chunk, as one sess.run does) vs the fp64 oracle driver, and the result
    must not depend on which other utterances share the batch."""
    topo = env["topology"].get("ModelL2LossWithoutDropoutLReluAttention")
    w = env["synthetic"].trained_like(topo, 23, seed=5)
    rng = np.random.default_rng(11)
    Ts = [25, 31, 100, 257, 600, 1300, 90]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in Ts]
    model = env["engine"].DeviceModel(w, topo, "cuda:0", precision=precision)
    vecs = env["engine"].Extractor(model, 25, 600).extract(mats)
    for m, v in zip(mats, vecs):
        ref = env["oracle"].embed_utterance(m, w, topo, 25, 600, np.float64)
        assert env["oracle"].rel_l2(v, ref) < TOL[precision], (m.shape[0], env["oracle"].rel_l2(v, ref))
    alone = env["engine"].Extractor(model, 25, 600).extract([mats[4]])[0]
    assert np.array_equal(alone, vecs[4])


@pytest.mark.parametrize("cls", ["ModelWithoutDropout", "ModelWithoutDropoutTdnn", "ModelWithoutDropoutPRelu", "ModelL2LossWithoutDropoutLRelu",
                                 "ModelL2LossWithoutDropoutReluHeInit", "ModelL2LossWithoutDropoutLReluAttention"])
def test_every_model_class_at_config2_batch_size(env, cls):
    """The variant classes in a full 262144-row batch of BASELINE configs[1]-shaped utterances (default precision): a few
    utterances spread over the batch against the fp64 oracle."""
    topo = env["topology"].get(cls)
    w = env["synthetic"].trained_like(topo, 23, seed=1)
    lens = env["synthetic"].utterance_lengths(850, 200, 400, 1234)
    rng = np.random.default_rng(8)
    mats = [(rng.standard_normal((int(T), 23)) * 3.0).astype(np.float32) for T in lens]
    model = env["engine"].DeviceModel(w, topo, "cuda:0")
    vecs = env["engine"].Extractor(model, 25, 10000).extract(mats)
    worst = 0.0
    for j in (0, 211, 424, 849):
        ref = env["oracle"].embed_utterance(mats[j], w, topo, 25, 10000, np.float64)
        worst = max(worst, env["oracle"].rel_l2(vecs[j], ref))
    assert worst < TOL["bf16x3"], (cls, worst)


def test_f16bf8_out_of_range_window_is_repeated_in_bf16x3(env, default_weights):
    """fp16 / bf8 end at 57344: a window in which some hidden activation exceeds that raises the status word of the f16bf8
    kernels, and the extractor repeats the window on the bf16x3 twin of the model -- the caller gets bf16x3 results, bit for
    bit; windows inside the range never touch the twin."""
    topo, w = default_weights
    oracle, engine = env["oracle"], env["engine"]
    rng = np.random.default_rng(23)
    base = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in (120, 333, 64)]
    model = engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8")
    assert model.f16bf8 and model.requested_precision == "f16bf8"
    ex = engine.Extractor(model, 25, 10000, accuracy_probe=False)      # (the run-time accuracy probe would build the twin)
    fine = ex.extract(base)
    assert ex.stats.get("fallback_windows", 0) == 0 and model._fallback is None
    for m, v in zip(base, fine):
        assert oracle.rel_l2(v, oracle.embed_utterance(m, w, topo, 25, 10000, np.float64)) < TOL["f16bf8"]
    wild = [base[0], base[1] * np.float32(1e6), base[2]]
    got = ex.extract(wild)
    assert ex.stats["fallback_windows"] == 1
    ref = engine.Extractor(engine.DeviceModel(w, topo, "cuda:0", precision="bf16x3"), 25, 10000).extract(wild)
    for a, b in zip(got, ref):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    again = ex.extract(base)                               # the status word is per window: the next one is f16bf8 again
    assert ex.stats["fallback_windows"] == 1
    for a, b in zip(again, fine):
        assert np.array_equal(a, b)


def test_f16bf8_falls_back_to_bf16x3_for_topologies_it_does_not_cover(env):
    """Attention pooling (and any hidden layer outside K in {1,3,5,7}, span <= 8) is not covered: the model then runs as bf16x3."""
    topo = env["topology"].get("ModelL2LossWithoutDropoutLReluAttention")
    w = env["synthetic"].trained_like(topo, 23, seed=3)
    model = env["engine"].DeviceModel(w, topo, "cuda:0", precision="f16bf8")
    assert not model.f16bf8 and model.precision == "bf16x3"


@pytest.mark.parametrize("feat", [30, 20, 40])
def test_f16bf8_with_a_first_layer_the_dedicated_kernel_does_not_take(env, feat):
    """30-dimensional MFCCs (the VoxCeleb recipes) with K = 5: 5 x 32 = 160 columns of im2col are beyond the first-layer kernel's 128, and
    20 dimensions are not a multiple of 8.  The model still runs its hidden layers in f16bf8 -- layer 0 on the general bf16x3 GEMM plus one
    encoding pass -- instead of falling back to bf16x3 as a whole; x-vectors against the fp64 oracle, through the extractor."""
    engine, oracle = env["engine"], env["oracle"]
    topo = env["topology"].get("ModelWithoutDropout")
    w = env["synthetic"].trained_like(topo, feat, seed=4)
    model = engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8")
    assert model.f16bf8 and model.first is None
    rng = np.random.default_rng(feat)
    mats = [(rng.standard_normal((t, feat)) * 3).astype(np.float32) for t in (300, 25, 77, 1000, 410)]
    ex = engine.Extractor(model, 25, 10000, accuracy_probe=False)
    got = ex.extract(mats)
    for m, v in zip(mats, got):
        assert oracle.rel_l2(v, oracle.embed_utterance(m, w, topo, 25, 10000, np.float64)) < TOL["f16bf8"]
    twin = engine.Extractor(engine.DeviceModel(w, topo, "cuda:0", precision="bf16x3"), 25, 10000).extract(mats)
    assert any(not np.array_equal(a, b) for a, b in zip(got, twin))          # (it did not quietly run as bf16x3)
