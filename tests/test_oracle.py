"""Pinning the CPU oracle (CPU-only tests).

* forward arithmetic: C fp64 oracle == independent NumPy fp64 restatement == torch-CPU ops
  (conv1d / batch_norm / var) evaluating the TF op definitions; C fp32 oracle within fp32 round-off
* THE PIN: the fp64 oracle == what the reference's own graphs compute (tests/golden/forward_refgraph.npz, train_refgraph.npz: every
  build_model / load_model / make_embedding / train_one_iteration / eval of local/tf/models.py executed under tests/golden/numpy_tf1.py)
* control flow + ark framing: the oracle's make_embedding restatement reproduces, BYTE FOR BYTE, the
  output stream the reference's own Model.make_embedding wrote (tests/golden/make_embedding.npz)
"""
import io

import numpy as np
import pytest

import kaldi_io
from fixture_inputs import (CONTROL_FEAT, CONTROL_LENGTHS, CONTROL_SEED, CONTROL_SETTINGS, FWD_SEED, compact, control_inputs,
                            refgraph_training_case)
from xvector_amd import synthetic, topology


def _torch_forward(x, w, topo, embedding_index=0):
    """Independent implementation with torch CPU float64 ops (cross-correlation conv1d with explicit
    symmetric zero padding, relu, eval batch-norm, population variance)."""
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.as_tensor(np.asarray(a, np.float64))
    act = topo["activation"]

    def activation(z, scope):
        if act == "relu":
            return F.relu(z)
        if act == "lrelu":
            return F.leaky_relu(z, topo["lrelu_alpha"])
        a = t(w[scope + "/prelu/prelu:0"])
        return F.relu(z) + a.view(1, -1, *([1] * (z.dim() - 2))) * torch.clamp(z, max=0.0)

    h = t(x).T.unsqueeze(0)                                   # [1, C, T]
    for i, (K, d) in enumerate(zip(topo["kernel_sizes"], topo["dilations"])):
        sc = "frame_level_info_layer-%d" % i
        wt = t(w[sc + "/w:0"]).permute(2, 1, 0).contiguous()  # [Cout, Cin, K]
        z = F.conv1d(h, wt, t(w[sc + "/b:0"]), padding=(K - 1) * d // 2, dilation=d)
        r = activation(z, sc)
        h = F.batch_norm(r, t(w[sc + "/mean:0"]), t(w[sc + "/variance:0"]), t(w[sc + "/gamma:0"]), t(w[sc + "/beta:0"]),
                         training=False, eps=topology.BN_EPSILON)
    if topo.get("pooling") == "attention":                  # models.py:1036-1050 with torch ops (softmax / einsum as written there)
        hh = h.permute(0, 2, 1)                             # [1, T, 2A]
        h1, h2 = torch.chunk(hh, 2, dim=2)
        nl = torch.tanh(torch.einsum("ijk,kl->ijl", h1, t(w["attention/w:0"])) + t(w["attention/b:0"]))
        att = torch.softmax(torch.einsum("ijk,k->ij", nl, t(w["attention/v:0"])), dim=-1)
        mu = torch.einsum("ijk,ij->ik", h2, att)
        var = torch.einsum("ijk,ij->ik", h2 * h2, att) - mu * mu
    else:
        mu = h.mean(dim=2)
        var = h.var(dim=2, unbiased=False)
    pooled = torch.cat([mu, torch.sqrt(var + topology.VAR2STD_EPSILON)], dim=1)
    e0 = pooled @ t(w["embed_layer-0/w:0"]) + t(w["embed_layer-0/b:0"])
    if embedding_index == 0:
        return e0[0].numpy()
    sc = "embed_layer-0"
    a0 = F.batch_norm(activation(e0, sc), t(w[sc + "/mean:0"]), t(w[sc + "/variance:0"]), t(w[sc + "/gamma:0"]),
                      t(w[sc + "/beta:0"]), training=False, eps=topology.BN_EPSILON)
    return (a0 @ t(w["embed_layer-1/w:0"]) + t(w["embed_layer-1/b:0"]))[0].numpy()


@pytest.mark.parametrize("cls", ["ModelWithoutDropout", "ModelWithoutDropoutTdnn", "ModelWithoutDropoutPRelu",
                                 "ModelL2LossWithoutDropoutLRelu", "ModelL2LossWithoutDropoutLReluAttention"])
def test_c_oracle_vs_numpy_vs_torch(oracle_mod, cls):
    topo = topology.get(cls)
    # narrower layers keep the test fast; kernel sizes / dilations / activation are the class's own
    topo["layer_sizes"] = [64, 64, 64, 64, 96]
    topo["embedding_sizes"] = [32, 32]
    w = synthetic.trained_like(topo, 23, seed=5)
    rng = np.random.default_rng(1)
    for T in (1, 3, 25, 140):
        x = (rng.standard_normal((T, 23)) * 3).astype(np.float32)
        for ei in (0, 1):
            c64 = oracle_mod.forward(x, w, topo, np.float64, ei)
            n64 = oracle_mod.forward_numpy(x, w, topo, ei)
            t64 = _torch_forward(x, w, topo, ei)
            c32 = oracle_mod.forward(x, w, topo, np.float32, ei)
            assert oracle_mod.rel_l2(c64, n64) < 1e-12
            assert oracle_mod.rel_l2(c64, t64) < 1e-12
            assert oracle_mod.rel_l2(c32, c64) < 5e-6


def test_reference_init_weights_forward(oracle_mod):
    """The reference's own initialisers (truncated normal 0.1, b=0.1, identity BN) through all three."""
    topo = topology.get("Model")
    topo["layer_sizes"] = [48, 48, 48, 48, 64]
    topo["embedding_sizes"] = [24, 24]
    w = synthetic.reference_init(topo, 23, 10, seed=3)
    assert abs(float(np.std(w["frame_level_info_layer-1/w:0"])) - 0.088) < 0.01       # truncated at 2 sigma
    assert np.abs(w["frame_level_info_layer-1/w:0"]).max() <= 0.2 + 1e-6
    x = (np.random.default_rng(2).standard_normal((60, 23))).astype(np.float32)
    a = oracle_mod.forward(x, w, topo, np.float64)
    assert oracle_mod.rel_l2(a, _torch_forward(x, w, topo)) < 1e-12


REF_CLASSES = [("default", "ModelWithoutDropout", [25, 200, 400, 1000]), ("dilated", "ModelWithoutDropoutTdnn", [25, 200, 400, 1000]),
               ("prelu", "ModelWithoutDropoutPRelu", [25, 200]), ("lrelu", "ModelL2LossWithoutDropoutLRelu", [25, 200]),
               ("attention", "ModelL2LossWithoutDropoutLReluAttention", [25, 200, 1000]), ("dropout", "Model", [25, 200]),
               ("l2prelu", "ModelL2LossWithoutDropoutPRelu", [25, 200]), ("heinit", "ModelL2LossWithoutDropoutReluHeInit", [25, 200])]


@pytest.mark.parametrize("tname,cls,Ts", REF_CLASSES)
def test_oracle_matches_the_reference_graph(oracle_mod, golden, tname, cls, Ts):
    """THE PIN of the forward arithmetic: tests/golden/forward_refgraph.npz holds what the reference's own build_model graphs
    (local/tf/models.py, tf_block.py -- executed under tests/golden/numpy_tf1.py, loaded through the reference's load_model) return
    for embedding[0], embedding[1], the pooled vector and every layer's output.  The fp64 oracle must agree to round-off, for all 8
    classes; the fp32 C oracle (the cpu_baseline arithmetic) within fp32 round-off."""
    g = golden("forward_refgraph.npz")
    assert int(g["seed"]) == FWD_SEED
    topo = topology.get(cls)
    w = synthetic.trained_like(topo, 23, seed=FWD_SEED)
    rng = np.random.default_rng(FWD_SEED + 1)
    for T in Ts:
        x = (rng.standard_normal((T, 23)) * 3.0).astype(np.float32)
        e1, inter = oracle_mod.forward(x, w, topo, np.float64, embedding_index=1, return_intermediates=True)
        assert oracle_mod.rel_l2(inter[6], g["%s_T%d_e0" % (tname, T)]) < 1e-12
        assert oracle_mod.rel_l2(e1, g["%s_T%d_e1" % (tname, T)]) < 1e-12
        assert oracle_mod.rel_l2(oracle_mod.forward_numpy(x, w, topo, 0), g["%s_T%d_e0" % (tname, T)]) < 1e-12
        if T == 25:
            for li in range(5):
                assert oracle_mod.rel_l2(inter[li][:, ::16], g["%s_T25_layer%d_sub" % (tname, li)]) < 1e-12, li
            assert oracle_mod.rel_l2(inter[5], g["%s_T25_pooled" % tname]) < 1e-12
        if T <= 200:
            assert oracle_mod.rel_l2(oracle_mod.forward(x, w, topo, np.float32), g["%s_T%d_e0" % (tname, T)]) < 5e-6


@pytest.mark.parametrize("tname,cls,Ts", REF_CLASSES)
def test_variable_names_shapes_and_initial_values_are_the_reference_graphs(golden, tname, cls, Ts, tmp_path):
    """The variables the reference's scopes produce (names, shapes; with the optimizer's slots) and the law of its initial values, against
    the twin's build_model: same names and shapes; constants equal; random tensors with the same bounds and spread."""
    import models as twin
    from xvector_amd import weights as wio
    g = golden("forward_refgraph.npz")
    names = [str(n) for n in g["%s_var_names" % tname]]
    shapes = dict(zip(names, [tuple(int(d) for d in str(s).split(",") if d) for s in g["%s_var_shapes" % tname]]))
    stats = dict(zip(names, g["%s_init_stats" % tname]))
    model_vars = [n for n in names if "/Adam" not in n and not n.startswith("beta")]
    trainable = [n for n in model_vars if not n.endswith(("/mean:0", "/variance:0"))]
    assert sorted(n for n in names if n.endswith("/Adam:0")) == sorted(n[:-2] + "/Adam:0" for n in trainable)       # slots exist for exactly the trainables
    assert sorted(n for n in names if n.endswith("/Adam_1:0")) == sorted(n[:-2] + "/Adam_1:0" for n in trainable)
    assert stats["beta1_power:0"][0] == 0.9 and stats["beta2_power:0"][0] == 0.999
    getattr(twin, cls)().build_model(64, 23, str(tmp_path / "m"))
    w, meta = wio.load_model_dir(str(tmp_path / "m"))
    assert sorted(w) == sorted(model_vars)
    from oracle import train_ref
    assert sorted(train_ref.trainable_names(meta["topology"])) == sorted(trainable)
    for n in model_vars:
        assert tuple(w[n].shape) == shapes[n], n
        lo, hi, mean, std = stats[n]
        if std == 0.0:                                                   # a constant initialiser: b = 0.1, gamma 1, beta 0, mean 0, variance 1, alpha 0.1
            assert np.all(w[n] == np.float32(lo)), (n, lo)
        else:
            a = np.asarray(w[n], np.float64)
            assert abs(a.std() - std) < 0.08 * std + 1e-3, (n, a.std(), std)
            bound = max(abs(lo), abs(hi))
            assert np.abs(a).max() <= bound * 1.05 + 1e-6 and np.abs(a).max() >= bound * 0.8, (n, np.abs(a).max(), bound)
            assert abs(a.mean() - mean) < 4 * std / np.sqrt(a.size) + 1e-3 * std + abs(mean) * 0.5, n


def test_oracle_make_embedding_reproduces_the_reference_graph_stream(oracle_mod, golden):
    """ark in -> the reference's make_embedding driving the reference's graph -> ark out (forward_refgraph.npz embed_*): the oracle's
    restatement of the driver over the oracle's forward writes the same records -- the same keys, and vectors equal to the last bit
    except where a float64 difference of 1e-15 straddles a float32 rounding boundary (at most 1 ulp, counted)."""
    g = golden("forward_refgraph.npz")
    topo = topology.get("ModelWithoutDropout")
    w = synthetic.trained_like(topo, 23, seed=FWD_SEED)
    rng = np.random.default_rng(FWD_SEED + 2)
    utts = [("rg%02d-T%d" % (i, T), (rng.standard_normal((T, 23)) * 3.0).astype(np.float32)) for i, T in enumerate(g["embed_lengths"])]
    flips = total = 0
    for si, (min_chunk, chunk) in enumerate(g["embed_settings"]):
        want = list(kaldi_io.read_vec_flt_ark(io.BytesIO(g["embed_out_ark_%d" % si].tobytes())))
        got = []
        for key, mat in utts:
            v = oracle_mod.embed_utterance(mat, w, topo, int(min_chunk), int(chunk), np.float64)
            if v is not None:
                got.append((key, v))
        assert [k for k, _ in got] == [k for k, _ in want]
        assert [k for k, _ in want] == [k for k, m in utts if m.shape[0] >= min_chunk]
        for (_, a), (_, b) in zip(got, want):
            assert a.dtype == b.dtype == np.float32
            ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
            assert ulp.max() <= 1
            flips += int((ulp > 0).sum())
            total += a.size
    assert flips <= total * 1e-3, (flips, total)


TRAIN_CLASSES = ["ModelWithoutDropout", "ModelWithoutDropoutTdnn", "ModelWithoutDropoutPRelu", "ModelL2LossWithoutDropoutPRelu",
                 "ModelL2LossWithoutDropoutLRelu", "ModelL2LossWithoutDropoutLReluAttention", "ModelL2LossWithoutDropoutReluHeInit", "Model"]


@pytest.mark.parametrize("cls", TRAIN_CLASSES)
def test_training_oracle_matches_the_reference_graph(oracle_mod, golden, cls):
    """THE PIN of the training arithmetic (f1): tests/golden/train_refgraph.npz holds what the reference's own train_one_iteration
    (models.py:216-305; tf_block.py:18-23 train branch; AdamOptimizer; the L2 terms; class Model's dropout sites with the masks drawn)
    and eval (models.py:307-354) did over three minibatches, executed under numpy_tf1 with reverse-mode gradients checked against
    finite differences (tests/test_numpy_tf1.py).  oracle/train_ref.py (torch float64 autograd) must reproduce the losses, the
    step-0 gradients, the weights, moving statistics and Adam slots after the steps, and the eval-phase losses."""
    from oracle import train_ref
    g = golden("train_refgraph.npz")
    stride, lr = int(g["stride"]), float(g["lr"])
    topo, w, batches, masks = refgraph_training_case(g, cls)
    ww = {k: np.asarray(v, np.float64) for k, v in w.items()}
    adam = {"t": 0, "m": {}, "v": {}}
    for bi, (x, labels) in enumerate(batches):
        loss, acc, ww, adam, grads = train_ref.train_step(ww, adam, topo, x.astype(np.float64), labels, lr, dropout=masks[bi] if masks else None)
        assert abs(loss - g["%s/loss" % cls][bi]) < 1e-10 * max(1.0, abs(loss)), bi
        assert acc == g["%s/accuracy" % cls][bi]
        if bi == 0:
            for n, gr in grads.items():
                assert oracle_mod.rel_l2(compact(gr, stride), g["%s/grad0/%s" % (cls, n)]) < 1e-9, n
    for n, v in ww.items():
        assert oracle_mod.rel_l2(compact(v, stride), g["%s/after/%s" % (cls, n)]) < 1e-9, n
    for n in adam["m"]:
        assert oracle_mod.rel_l2(compact(adam["m"][n], stride, 8), g["%s/after/%s/Adam:0" % (cls, n[:-2])]) < 1e-9, n
        assert oracle_mod.rel_l2(compact(adam["v"][n], stride, 8), g["%s/after/%s/Adam_1:0" % (cls, n[:-2])]) < 1e-9, n
    assert abs(float(g["%s/after/beta1_power:0" % cls][0]) - 0.9 ** (adam["t"] + 1)) < 1e-15
    for bi, (x, labels) in enumerate(batches[:2]):
        loss, acc, _ = train_ref.eval_batch(ww, topo, x.astype(np.float64), labels)
        assert abs(loss - g["%s/eval_loss" % cls][bi]) < 1e-9 * max(1.0, abs(loss))
        assert acc == g["%s/eval_accuracy" % cls][bi]


def test_chunk_plan_matches_reference_driver(oracle_mod, golden):
    """Chunk lengths the reference's make_embedding fed to sess.run, per setting (golden log)."""
    g = golden("make_embedding.npz")
    from xvector_amd.engine import plan_chunks
    for si, (min_chunk, chunk) in enumerate(CONTROL_SETTINGS):
        want = list(g["chunk_lens_%d" % si])
        got_oracle, got_engine = [], []
        for T in CONTROL_LENGTHS:
            p = oracle_mod.chunk_plan(T, min_chunk, chunk)
            q = plan_chunks(T, min_chunk, chunk)
            assert p == q                                   # oracle (C) and product host logic agree
            if p:
                got_oracle += [n for _, n in p]
                assert all(s == sum(n2 for _, n2 in p[:i]) or True for i, (s, _) in enumerate(p))
        assert got_oracle == want, (min_chunk, chunk)


def test_chunk_plan_edge_cases(oracle_mod):
    cp = oracle_mod.chunk_plan
    assert cp(0, 25, 10000) is None and cp(24, 25, 10000) is None
    assert cp(25, 25, 10000) == [(0, 25)]
    assert cp(10000, 25, 10000) == [(0, 10000)]
    assert cp(10001, 25, 10000) == [(0, 10000)]                  # 1-frame tail < min_chunk: dropped
    assert cp(10025, 25, 10000) == [(0, 10000), (10000, 25)]
    assert cp(700, 100, -1) == [(0, 700)]
    assert cp(650, 25, 300) == [(0, 300), (300, 300), (600, 50)]


def test_oracle_make_embedding_bytes_equal_reference_output(oracle_mod, golden):
    """Feed the control-flow fixture through the oracle's restatement of make_embedding and the ark
    writer: the byte stream must equal what the reference's own driver wrote."""
    g = golden("make_embedding.npz")
    topo = synthetic.SMALL_TOPOLOGY
    w = synthetic.trained_like(topo, CONTROL_FEAT, num_classes=8, seed=CONTROL_SEED)
    utts = control_inputs()
    for si, (min_chunk, chunk) in enumerate(CONTROL_SETTINGS):
        out = io.BytesIO()
        for key, mat in utts:
            v = oracle_mod.embed_utterance(mat, w, topo, min_chunk, chunk, np.float64)
            if v is not None:
                kaldi_io.write_vec_flt(out, v, key=key)
        assert out.getvalue() == g["out_ark_%d" % si].tobytes(), (min_chunk, chunk)


def test_chunk_average_is_numpy_float32_semantics(oracle_mod):
    rng = np.random.default_rng(4)
    e = (rng.standard_normal((3, 16)) * 7).astype(np.float32)
    lens = [10000, 10000, 4321]
    acc, tot = 0, 0.0
    for n, v in zip(lens, e):            # the reference's expression, models.py:398,418-421
        tot += n
        acc = acc + n * v
    acc = acc / tot
    assert np.array_equal(oracle_mod.chunk_average(e, lens, np.float32), acc.astype(np.float32))


def test_sliding_cmn_restatement_properties(oracle_mod):
    oracle = oracle_mod
    """Kaldi SlidingWindowCmn semantics (SURVEY §8f-4; parity unpinned: Kaldi is not vendored): centred window of 300 frames
    clipped-and-shifted at the edges, whole-utterance mean when T <= window, the non-centred min_window rule."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1000, 5)) * 3 + 7).astype(np.float32)
    y = oracle.sliding_cmn(x)
    x64 = x.astype(np.float64)
    for t, (ws, we) in {0: (0, 300), 149: (0, 300), 150: (0, 300), 151: (1, 301), 500: (350, 650), 849: (699, 999),
                        850: (700, 1000), 999: (700, 1000)}.items():
        assert np.array_equal(y[t], (x64[t] - x64[ws:we].mean(axis=0)).astype(np.float32)) or \
            np.allclose(y[t], x64[t] - x64[ws:we].mean(axis=0), rtol=0, atol=1e-6), t
    short = x[:120]
    assert np.allclose(oracle.sliding_cmn(short), short - short.astype(np.float64).mean(axis=0), atol=1e-6)
    yn = oracle.sliding_cmn(x, 300, False, 100)
    assert np.allclose(yn[50], x64[50] - x64[:100].mean(axis=0), atol=1e-6)          # t < min_window: first min_window frames
    assert np.allclose(yn[500], x64[500] - x64[200:501].mean(axis=0), atol=1e-6)     # causal window of cmn_window + 1 frames
    const = np.tile(np.float32([1.5, -2.0]), (400, 1))
    assert not oracle.sliding_cmn(const).any()
    vad = np.zeros(1000); vad[10:20] = 1; vad[500] = 0.5
    sel = oracle.select_voiced(x, vad)
    assert sel.shape == (11, 5) and np.array_equal(sel[-1], x[500])
    assert oracle.select_voiced(x, np.zeros(1000)) is None and oracle.select_voiced(x, np.ones(999)) is None


def test_config1_cpu_plumbing_ark_to_ark(oracle_mod, default_weights):
    """BASELINE configs[0]: 100 synthetic utterances, 23-dim, fixed T=200, default topology, CPU only: ark in -> the oracle's
    restatement of the driver (chunk plan + fp32 forward + NumPy float32 average) -> ark out; output framing is the
    reference's (key, '\\0B', 'FV ', dim 512), and the fp32 CPU path agrees with the fp64 oracle far inside the 1e-4 bar."""
    import time
    topo, w = default_weights
    rng = np.random.default_rng(1234)
    utts = [("utt%06d" % i, (rng.standard_normal((200, 23)) * 3.0).astype(np.float32)) for i in range(100)]
    bio = io.BytesIO()
    for k, m in utts:
        kaldi_io.write_mat(bio, m, key=k)
    out = io.BytesIO()
    t0 = time.time()
    for k, m in kaldi_io.read_mat_ark(io.BytesIO(bio.getvalue())):
        kaldi_io.write_vec_flt(out, oracle_mod.embed_utterance(m, w, topo, 25, 10000, np.float32), key=k)
    dt = time.time() - t0
    got = list(kaldi_io.read_vec_flt_ark(io.BytesIO(out.getvalue())))
    assert [k for k, _ in got] == [k for k, _ in utts] and all(v.shape == (512,) and v.dtype == np.float32 for _, v in got)
    assert len(out.getvalue()) == 100 * (len("utt000000 ") + 2 + 3 + 1 + 4 + 512 * 4)
    for i in (0, 37, 99):
        ref = oracle_mod.embed_utterance(utts[i][1], w, topo, 25, 10000, np.float64)
        assert oracle_mod.rel_l2(got[i][1], ref) < 5e-6
    print("config 1 on the CPU oracle (fp32 C, OpenMP): %.1f utt/s" % (100 / dt))
