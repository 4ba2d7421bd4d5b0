"""Feature front-end on the GPU vs the NumPy restatement of Kaldi's SlidingWindowCmn / select-voiced-frames
(oracle/oracle.py).  Both sides form the window mean in float64 and round the difference to float32 once; only the order
of the float64 additions differs, so the results agree bit for bit except where that last-place difference flips the final
rounding: asserted as <= 1 float32 ulp everywhere and identical on > 99.9 % of the elements."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from oracle import oracle
    from xvector_amd import frontend, hiplib
    hiplib.require_gpu()
    return dict(torch=torch, oracle=oracle, frontend=frontend, hiplib=hiplib)


def _close(got, ref):
    assert got.shape == ref.shape and got.dtype == np.float32
    ulp = np.spacing(np.maximum(np.abs(ref), np.float32(1e-3)).astype(np.float32))
    assert np.all(np.abs(got.astype(np.float64) - ref.astype(np.float64)) <= ulp)
    assert np.mean(got == ref) > 0.999


@pytest.mark.parametrize("center,window,F", [(True, 300, 23), (False, 300, 23), (True, 61, 40), (False, 600, 5)])
def test_cmn_select_matches_oracle(env, center, window, F):
    oracle = env["oracle"]
    rng = np.random.default_rng(window + F)
    lens = [1, 2, 50, 299, 300, 301, 1000, 2500, 64, 65]
    mats = [(rng.standard_normal((t, F)) * 4 + 10 * rng.standard_normal(F)).astype(np.float32) for t in lens]
    vads = [(rng.random(t) < 0.7).astype(np.float32) for t in lens]
    vads[0][:] = 1.0
    vads[2] = None                                        # no VAD for this utterance: every frame is kept
    fe = env["frontend"].FrontEnd("cuda:0", cmn_window=window, center=center, min_window=100)
    outs = fe.apply(mats, vads)
    for m, v, got in zip(mats, vads, outs):
        ref = oracle.sliding_cmn(m, window, center, 100)
        if v is not None:
            ref = oracle.select_voiced(ref, v)
        if ref is None:                                   # nothing voiced in a very short utterance: dropped by both
            assert got is None
            continue
        _close(got, ref)
    # no VAD at all
    outs = fe.apply(mats[:4])
    for m, got in zip(mats[:4], outs):
        _close(got, oracle.sliding_cmn(m, window, center, 100))


def test_frontend_drops_what_select_voiced_frames_drops(env):
    rng = np.random.default_rng(0)
    mats = [(rng.standard_normal((t, 23))).astype(np.float32) for t in (100, 100, 100, 0)]
    vads = [np.ones(100, np.float32), np.zeros(100, np.float32), np.ones(99, np.float32), np.zeros(0, np.float32)]
    fe = env["frontend"].FrontEnd("cuda:0")
    outs = fe.apply(mats, vads)
    assert outs[0].shape == (100, 23) and outs[1] is None and outs[2] is None and outs[3] is None
    assert env["oracle"].select_voiced(mats[1], vads[1]) is None and env["oracle"].select_voiced(mats[2], vads[2]) is None
    assert fe.stats["dropped"] == 3 and fe.stats["frames_out"] == 100


def test_scatter_into_a_packed_batch(env):
    """The kernel's dst_row indirection writes straight into a packed [R, in_dim] batch: untouched rows / columns keep their
    contents (gap rows and the 24th padding column stay zero)."""
    torch, hiplib, oracle = env["torch"], env["hiplib"], env["oracle"]
    rng = np.random.default_rng(3)
    T, F = 400, 23
    m = (rng.standard_normal((T, F)) * 3).astype(np.float32)
    x = torch.from_numpy(m).cuda()
    y = torch.zeros((T + 16, 24), device="cuda")
    dst = np.full(T, -1, np.int32)
    dst[100:300] = np.arange(8, 208)                      # one 200-frame chunk placed at row 8
    hiplib.cmn_sliding_scatter(x, torch.zeros(1, dtype=torch.int32, device="cuda"), torch.full((1,), T, dtype=torch.int32, device="cuda"),
                               1, T, 300, True, 100, torch.from_numpy(dst).cuda(), y)
    got = y.cpu().numpy()
    ref = oracle.sliding_cmn(m)
    _close(got[8:208, :F], ref[100:300])
    assert not got[:8].any() and not got[208:].any() and not got[:, F:].any()


def test_cli_raw_features_plus_vad_to_speaker_xvectors(env, tmp_path):
    """extract_xvectors.sh without Kaldi binaries: raw features (scp) + VAD (ark) -> extract_embedding.py --cmn-window 300
    --vad-rspecifier ... -> xvector.ark/scp -> speaker_mean.py; every x-vector against the oracle pipeline
    sliding_cmn -> select_voiced -> embed_utterance (fp64), dropped utterances as select-voiced-frames / make_embedding
    drop them."""
    import sys, os
    import kaldi_io
    import models
    import extract_embedding as ee
    import speaker_mean
    from xvector_amd import synthetic
    oracle = env["oracle"]
    topo = synthetic.SMALL_TOPOLOGY
    w = synthetic.trained_like(topo, 5, num_classes=8, seed=11)
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=8, feat_dim=5), mdir, None)
    rng = np.random.default_rng(1)
    lens = [400, 60, 250, 900, 120]
    utts = [("spkA-u%d" % i if i < 3 else "spkB-u%d" % i, (rng.standard_normal((T, 5)) * 3 + 5).astype(np.float32))
            for i, T in enumerate(lens)]
    vads = {k: (rng.random(m.shape[0]) < 0.8).astype(np.float32) for k, m in utts}
    vads[utts[1][0]][:] = 0                                       # nothing voiced -> dropped
    vads[utts[4][0]] = (np.arange(120) < 20).astype(np.float32)   # 20 voiced frames < min_chunk_size 25 -> rejected later
    feats_ark, feats_scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    with kaldi_io.TableWriter(feats_ark, feats_scp) as tw:
        for k, m in utts:
            kaldi_io.write_mat(tw, m, key=k)
    vad_ark = str(tmp_path / "vad.ark")
    with open(vad_ark, "wb") as f:
        for k, _ in utts:
            kaldi_io.write_vec_flt(f, vads[k], key=k)
    ark, scp = str(tmp_path / "xvector.ark"), str(tmp_path / "xvector.scp")
    ee.main(["--min-chunk-size", "25", "--chunk-size", "300", "--feature-rspecifier", "scp:" + feats_scp,
             "--vector-wspecifier", "ark,scp:%s,%s" % (ark, scp), "--model-dir", mdir, "--cmn-window", "300",
             "--vad-rspecifier", "ark:" + vad_ark])
    got = dict(kaldi_io.read_vec_flt_scp(scp))
    assert list(got) == [utts[0][0], utts[2][0], utts[3][0]]
    for k, m in utts:
        sel = oracle.select_voiced(oracle.sliding_cmn(m), vads[k])
        ref = None if sel is None else oracle.embed_utterance(sel, w, topo, 25, 300, np.float64)
        assert (ref is None) == (k not in got), k
        if ref is not None:
            assert oracle.rel_l2(got[k], ref) < 5e-5, k
    spk2utt = tmp_path / "spk2utt"
    spk2utt.write_text("spkA %s %s %s\nspkB %s %s\nspkC ghost\n" % tuple(k for k, _ in utts))
    sark, sscp, nutt = str(tmp_path / "spk.ark"), str(tmp_path / "spk.scp"), str(tmp_path / "num_utts.ark")
    speaker_mean.main([str(spk2utt), scp, sark, sscp, nutt])
    spk = dict(kaldi_io.read_vec_flt_scp(sscp))
    assert list(spk) == ["spkA", "spkB"] and open(nutt).read() == "spkA 2\nspkB 1\n"
    np.testing.assert_allclose(spk["spkA"], (got[utts[0][0]] + got[utts[2][0]]) / np.float32(2), rtol=1e-6)
    assert np.array_equal(spk["spkB"], got[utts[3][0]])


def test_device_front_end_path_equals_host_round_trip(tmp_path):
    """Extractor.submit_raw (CMN + voiced-frame selection scattered straight into the packed batches on the device) gives
    bit-identical x-vectors to FrontEnd.apply + Extractor.extract (selected features through the host), across chunking,
    small batches that split utterances, VADs that drop utterances, missing VADs and an all-voiced window."""
    from xvector_amd import engine, frontend, synthetic, topology
    if engine._host_lib() is None:
        pytest.skip("libxvector_host.so disabled: make_embedding then takes the host round trip")
    topo = topology.get("ModelWithoutDropout")
    w = synthetic.trained_like(topo, 23, seed=3)
    model = engine.DeviceModel(w, topo, "cuda:0")
    rng = np.random.default_rng(12)
    Ts = [300, 40, 1, 0, 777, 25, 1300, 90, 260, 33]
    mats = [(rng.standard_normal((t, 23)) * 3 + 1.5).astype(np.float32) for t in Ts]
    vads = [(rng.random(t) < 0.7).astype(np.float32) for t in Ts]
    vads[1] = np.zeros(40, np.float32)              # nothing voiced -> dropped
    vads[4] = vads[4][:-1]                          # length mismatch -> dropped
    vads[7] = None                                  # no VAD for this key: every frame voiced
    for vv, (mn, cs, rows) in ((vads, (25, 200, 262144)), (vads, (10, -1, 512)), (None, (25, 300, 700)), (vads, (25, 10000, 262144))):
        fe = frontend.FrontEnd("cuda:0", 300, True)
        sel = fe.apply(mats, vv)
        ex = engine.Extractor(model, mn, cs, max_batch_rows=rows)
        keep = [i for i, m in enumerate(sel) if m is not None]
        want = ex.extract([sel[i] for i in keep])
        ex2 = engine.Extractor(model, mn, cs, max_batch_rows=rows)
        handle, lens, dropped = ex2.submit_raw(mats, vv, 300, True)
        got = ex2.finish(handle)
        assert [i for i in range(len(mats)) if not dropped[i]] == keep or vv is None
        assert [int(lens[i]) for i in keep] == [sel[i].shape[0] for i in keep]
        for j, i in enumerate(keep):
            assert (want[j] is None) == (got[i] is None), (i, mn, cs)
            if want[j] is not None:
                assert np.array_equal(want[j], got[i]), (i, mn, cs, rows)
        for i in range(len(mats)):
            if i not in keep:
                assert got[i] is None


def test_cli_on_a_kaldi_shaped_data_directory(env, tmp_path):
    """What a Kaldi data directory really holds: COMPRESSED feature matrices (make_mfcc.sh's default) spread over several arks, a
    feats.scp that lists a subset of them (utterances removed by a filter), a vad.scp over ALL utterances.  The CLI with the device
    front-end reads it through the native paths (xv_ark_decode_cm, gap-tolerant tables) and writes byte for byte the x-vector ark it
    writes for the same matrices stored as plain float matrices in one ark with exactly matching tables."""
    import kaldi_io
    import models
    import extract_embedding as ee
    from fixture_inputs import encode_cm_record
    from xvector_amd import synthetic
    topo = synthetic.SMALL_TOPOLOGY
    w = synthetic.trained_like(topo, 23, num_classes=8, seed=12)
    mdir = str(tmp_path / "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=8, feat_dim=23), mdir, None)
    rng = np.random.default_rng(2)
    n = 240
    keys = ["spk%02d-utt%04d" % (i % 7, i) for i in range(n)]
    recs = [encode_cm_record(k, (rng.standard_normal((int(rng.integers(30, 400)), 23)) * 3 + 2).astype(np.float32)) for k in keys]
    mats = [m for _, m in kaldi_io.read_mat_ark(__import__("io").BytesIO(b"".join(recs)))]       # what the compressed bytes decode to
    vads = [(rng.random(m.shape[0]) < 0.8).astype(np.float32) for m in mats]
    # (a) the Kaldi-shaped directory: three compressed arks, scp offsets, 85 % of the utterances listed; vad.scp lists all
    lines = []
    for part in range(3):
        path = str(tmp_path / ("raw_mfcc.%d.ark" % (part + 1)))
        with open(path, "wb") as f:
            for i in range(part * 80, (part + 1) * 80):
                lines.append("%s %s:%d" % (keys[i], path, f.tell() + len(keys[i]) + 1))
                f.write(recs[i])
    keep = [i for i in range(n) if rng.random() < 0.85]
    feats_scp = str(tmp_path / "feats.scp")
    open(feats_scp, "wt").write("\n".join(lines[i] for i in keep) + "\n")
    vad_ark, vad_scp = str(tmp_path / "vad.ark"), str(tmp_path / "vad.scp")
    with kaldi_io.TableWriter(vad_ark, vad_scp) as tw:
        for k, v in zip(keys, vads):
            kaldi_io.write_vec_flt(tw, v, key=k)
    # (b) the same utterances as plain float matrices in one ark, tables that list exactly them
    pf_ark, pf_scp = str(tmp_path / "plain.ark"), str(tmp_path / "plain.scp")
    with kaldi_io.TableWriter(pf_ark, pf_scp) as tw:
        for i in keep:
            kaldi_io.write_mat(tw, mats[i], key=keys[i])
    pv_ark, pv_scp = str(tmp_path / "pvad.ark"), str(tmp_path / "pvad.scp")
    with kaldi_io.TableWriter(pv_ark, pv_scp) as tw:
        for i in keep:
            kaldi_io.write_vec_flt(tw, vads[i], key=keys[i])
    out = {}
    for tag, fs, vs in (("kaldi", feats_scp, vad_scp), ("plain", pf_scp, pv_scp)):
        ark = str(tmp_path / (tag + "_xvector.ark"))
        ee.main(["--min-chunk-size", "25", "--chunk-size", "300", "--feature-rspecifier", "scp:" + fs, "--vector-wspecifier", "ark:" + ark,
                 "--model-dir", mdir, "--cmn-window", "300", "--vad-rspecifier", "scp:" + vs])
        out[tag] = open(ark, "rb").read()
    assert len(out["plain"]) > 1000 and out["kaldi"] == out["plain"]
    got = [k for k, _ in kaldi_io.read_vec_flt_ark(__import__("io").BytesIO(out["kaldi"]))]
    assert got == [keys[i] for i in keep if vads[i].sum() >= 25]


def test_vad_tables_in_any_order_give_the_same_vectors(tmp_path):
    """Model.make_embedding with the device front-end takes the VAD vectors in RUNS while the VAD table's keys arrive in the order of the
    features (frontend.VadRuns), steps over entries the features do not have, and falls back to one vector at a time when a key is
    out of order: a VAD table in feature order, one with extra keys, a shuffled one and a per-key iterator all write the same bytes;
    a table that lacks keys drops exactly those utterances."""
    import io
    import logging
    import kaldi_io
    import models
    from xvector_amd import synthetic, topology, weights as wio
    topo = topology.get("ModelWithoutDropout")
    w = synthetic.trained_like(topo, 23, seed=2)
    mdir = str(tmp_path / "m")
    wio.save_model_dir(mdir, w, topo, "ModelWithoutDropout", 8, 23)
    rng = np.random.default_rng(21)
    n = 70
    keys = ["utt%03d" % i for i in range(n)]
    mats = [(rng.standard_normal((int(rng.integers(60, 400)), 23)) * 3).astype(np.float32) for _ in range(n)]
    vads = [(rng.random(m.shape[0]) < 0.75).astype(np.float32) for m in mats]
    with kaldi_io.TableWriter(str(tmp_path / "f.ark"), str(tmp_path / "f.scp")) as t:
        for k, m in zip(keys, mats):
            kaldi_io.write_mat(t, m, key=k)
    extra = [("zzz%02d" % i, np.ones(17, np.float32)) for i in range(5)]

    def vad_table(name, order, with_extra=False, drop=()):
        entries = [(keys[i], vads[i]) for i in order if keys[i] not in drop]
        if with_extra:                                       # entries the features do not have, in between and at both ends
            entries = [("aaa", np.ones(9, np.float32))] + entries[:20] + extra[:2] + entries[20:51] + extra[2:] + entries[51:]
        with kaldi_io.TableWriter(str(tmp_path / (name + ".ark")), str(tmp_path / (name + ".scp"))) as tv:
            for k, v in entries:
                kaldi_io.write_vec_flt(tv, v, key=k)
        return str(tmp_path / (name + ".scp"))

    log = logging.getLogger("vad-order")
    log.addHandler(logging.NullHandler())

    def run(vad_source):
        out = io.BytesIO()
        models.ModelWithoutDropout().make_embedding(kaldi_io.MatScp(str(tmp_path / "f.scp")), out, mdir, 25, 10000, True, log,
                                                    vad_stream=vad_source, cmn_window=300)
        return out.getvalue()

    in_order = run(kaldi_io.VecScp(vad_table("v0", range(n))))
    got = dict(kaldi_io.read_vec_flt_ark(io.BytesIO(in_order)))
    assert list(got) == keys
    assert run(kaldi_io.VecScp(vad_table("v1", range(n), with_extra=True))) == in_order
    shuffled = list(rng.permutation(n))
    assert run(kaldi_io.VecScp(vad_table("v2", shuffled))) == in_order
    assert run(open(str(tmp_path / "v0.ark"), "rb")) == in_order                               # the VAD ark as a stream
    assert run(iter(list(zip(keys, vads)))) == in_order                                          # (key, vector) pairs
    missing = {keys[3], keys[40], keys[69]}
    part = dict(kaldi_io.read_vec_flt_ark(io.BytesIO(run(kaldi_io.VecScp(vad_table("v3", range(n), drop=missing))))))
    assert list(part) == [k for k in keys if k not in missing]
    assert all(np.array_equal(part[k], got[k]) for k in part)
