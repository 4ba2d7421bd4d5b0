"""GPU parity of the Toom-Cook F(2, K) form of the wide-context layers (xv_tdnn_layer_toom_f32, csrc/xv_toom.hip) through the C ABI
against the fp64 oracle of the DIRECT K-tap contraction (local/tf/models.py:60-65): the transform must not be visible beyond
rounding.  Tolerance: TOL_GEMM of test_gpu_kernels.py (2e-6 relative L2 per layer; the direct fp32 kernel sits at ~3e-7, this
form at ~1e-6)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_GEMM = 2e-6


@pytest.fixture(scope="module")
def env(oracle_mod):
    import torch
    from xvector_amd import engine, hiplib
    hiplib.require_gpu()
    return dict(torch=torch, hiplib=hiplib, engine=engine, oracle=oracle_mod, dev=torch.device("cuda:0"))


def _rand_bn(rng, c):
    return ((1 + 0.1 * rng.standard_normal(c)).astype(np.float32), (0.1 * rng.standard_normal(c)).astype(np.float32),
            (0.2 * rng.standard_normal(c)).astype(np.float32), np.exp(0.2 * rng.standard_normal(c)).astype(np.float32))


def _run(env, mats, w, b, bn, act, alpha, align=8, direct=False, dil=1):
    torch, hiplib, engine, dev = env["torch"], env["hiplib"], env["engine"], env["dev"]
    K, cin, cout = w.shape
    layout = engine.BatchLayout([m.shape[0] for m in mats], (K - 1) // 2 * dil, align)
    host = np.zeros((layout.rows, cin), np.float32)
    layout.pack(mats, host)
    x = torch.from_numpy(host).to(dev)
    rv = torch.from_numpy(layout.row_valid()).to(dev)
    wd = torch.from_numpy(np.ascontiguousarray(w)).to(dev)
    wp = hiplib.pack_weights(wd.reshape(K * cin, cout)) if direct else hiplib.pack_weights_toom(wd)
    scale, shift = hiplib.fold_bn(*(torch.from_numpy(a).to(dev) for a in bn), 1e-3)
    al = None if alpha is None else torch.from_numpy(np.atleast_1d(alpha).astype(np.float32)).to(dev)
    y = torch.full((layout.rows, cout), float("nan"), dtype=torch.float32, device=dev)
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]
    hiplib.tdnn_layer(x, wp, torch.from_numpy(b).to(dev), scale, shift, code, al, K, dil, rv, y)
    torch.cuda.synchronize()
    yh = y.cpu().numpy()
    return [yh[s:s + n] for s, n in zip(layout.row_start, layout.row_len)], yh, layout


@pytest.mark.parametrize("cin,cout,K,act,dil", [
    (512, 512, 5, "relu", 1),       # layer 1 of the default topology
    (512, 512, 7, "relu", 1),       # layer 2
    (64, 48, 5, "prelu", 1),       # ragged column tile, two slabs, PReLU epilogue
    (96, 200, 7, "lrelu", 1),       # odd slab count, two column tiles
    (32, 4, 5, "none", 1),       # one slab (odd slab count: the loop's tail), a 4-column layer
    (512, 512, 3, "relu", 2),       # layer 1 of the dilated class (models.py:545-548,579-585): F(2, 3) over every 2nd row
    (512, 512, 3, "relu", 3),       # its layer 2: every 3rd row, chunks on multiples of 24 rows
    (96, 200, 3, "prelu", 1),       # F(2, 3) undilated, odd slab count, two column tiles
    (64, 48, 5, "lrelu", 2),        # the wide kernels take a dilation the same way
    (32, 36, 7, "none", 3),
    (64, 64, 3, "relu", 8),         # the largest dilation the entry point takes
])
def test_toom_layer_matches_oracle(env, cin, cout, K, act, dil):
    oracle = env["oracle"]
    rng = np.random.default_rng(cin * 1000 + cout + K * 7 + 31 * dil)
    lens = [25, 1, 130, 257, 64, 3, 2, 127]          # chunks shorter than the halo, odd lengths, spanning tile boundaries
    mats = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in lens]
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    alpha = None
    if act == "lrelu":
        alpha = np.array([0.2], np.float32)
    elif act == "prelu":
        alpha = (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
    import math
    outs, yh, layout = _run(env, mats, w, b, bn, act, alpha, align=math.lcm(8, 2 * dil), dil=dil)
    for m, got in zip(mats, outs):
        ref = oracle.tdnn_layer(m, w, b, bn, act, alpha, dil, np.float64)
        assert np.isfinite(got).all()
        assert oracle.rel_l2(got, ref) < TOL_GEMM, (m.shape, oracle.rel_l2(got, ref))
    valid = layout.row_valid().astype(bool)
    assert (yh[~valid] == 0).all()                   # gap rows are written as exact zeros


def test_toom_bits_do_not_depend_on_batch_neighbours(env):
    """Row pairs sit on even global rows and chunks start on rows that are multiples of 8: a chunk alone and the same chunk in a
    26 k-row batch (other tiles, other neighbours, a neighbour's first row right behind its gap) come out bit-identical."""
    import math
    oracle = env["oracle"]
    rng = np.random.default_rng(11)
    for K, dil in ((5, 1), (7, 1), (3, 1), (3, 2), (3, 3), (5, 2)):
        cin, cout = 64, 128
        align = math.lcm(8, 2 * dil)                 # what DeviceModel lays batches out with for such a layer
        w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
        b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        bn = _rand_bn(rng, cout)
        # lengths chosen so that some chunks end 3 / 4 rows before the next one starts (T % 8 in {5, 4}) and some on odd rows
        probe = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in (133, 28, 257, 5, 300)]
        alone = [_run(env, [m], w, b, bn, "relu", None, align=align, dil=dil)[0][0] for m in probe]
        crowd = [(rng.standard_normal((int(t), cin)) * 2).astype(np.float32) for t in rng.integers(1, 400, 120)]
        mixed = crowd[:40] + [probe[0]] + crowd[40:80] + probe[1:3] + crowd[80:] + probe[3:]
        outs, _, _ = _run(env, mixed, w, b, bn, "relu", None, align=align, dil=dil)
        got = [outs[40], outs[81], outs[82], outs[-2], outs[-1]]
        for m, a, g in zip(probe, alone, got):
            assert np.array_equal(a, g), (K, dil)
            assert oracle.rel_l2(a, oracle.tdnn_layer(m, w, b, bn, "relu", None, dil, np.float64)) < TOL_GEMM


def test_toom_unsupported_shapes_are_refused(env):
    hiplib, torch, dev = env["hiplib"], env["torch"], env["dev"]
    assert hiplib.toom_supported(5, 1, 512, 512) and hiplib.toom_supported(7, 1, 32, 4) and hiplib.toom_supported(3, 3, 512, 512)
    for K, d, cin, cout in ((3, 9, 512, 512), (5, 0, 512, 512), (5, 1, 24, 512), (7, 1, 512, 510), (1, 1, 512, 512), (9, 1, 64, 64)):
        assert not hiplib.toom_supported(K, d, cin, cout)
    lib = hiplib.load()
    assert lib.xv_packed_weights_toom_f32_floats(5, 24, 512) == 0
    w = torch.zeros((5, 24, 512), device=dev)
    with pytest.raises(AssertionError):
        hiplib.pack_weights_toom(w)


@pytest.mark.parametrize("feat,K,cout,act", [(23, 5, 512, "relu"), (30, 5, 64, "prelu"), (40, 3, 200, "lrelu"), (13, 7, 32, "none")])
def test_first_layer_rows_form_matches_oracle(env, feat, K, cout, act):
    """xv_tdnn_layer_rows_f32: layer 0 as a K = 1 GEMM over the overlapping windows of the packed feature rows, against the fp64
    oracle of the K-tap contraction -- on a small batch (64-row tiles, register-staged kernel) and on a 26 k-row batch (DMA-fed
    kernel, rows before the buffer and past its end through the descriptor's range check); the chunks common to both must come
    out bit-identical, and chunks of 1-3 frames at both ends of the batch exercise the buffer bounds."""
    torch, hiplib, engine, dev, oracle = env["torch"], env["hiplib"], env["engine"], env["dev"], env["oracle"]
    rng = np.random.default_rng(feat * 100 + K)
    ldx = (feat + 3) // 4 * 4
    w = (rng.standard_normal((K, feat, cout)) / np.sqrt(K * feat)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    alpha = np.array([0.2], np.float32) if act == "lrelu" else (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32) if act == "prelu" else None
    wpad = np.zeros((K, ldx, cout), np.float32)
    wpad[:, :feat] = w
    wp = hiplib.pack_weights_rows(torch.from_numpy(wpad).to(dev), ldx)
    scale, shift = hiplib.fold_bn(*(torch.from_numpy(a).to(dev) for a in bn), 1e-3)
    al = None if alpha is None else torch.from_numpy(alpha).to(dev)
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]

    def run(mats):
        layout = engine.BatchLayout([m.shape[0] for m in mats], (K - 1) // 2, 8)
        host = np.zeros((layout.rows, ldx), np.float32)
        layout.pack(mats, host)
        x = torch.from_numpy(host).to(dev)
        rv = torch.from_numpy(layout.row_valid()).to(dev)
        y = torch.full((layout.rows, cout), float("nan"), dtype=torch.float32, device=dev)
        hiplib.tdnn_layer(x, wp, torch.from_numpy(b).to(dev), scale, shift, code, al, K, 1, rv, y)
        torch.cuda.synchronize()
        yh = y.cpu().numpy()
        assert (yh[~layout.row_valid().astype(bool)] == 0).all()
        return [yh[s:s + n] for s, n in zip(layout.row_start, layout.row_len)]
    small = [(rng.standard_normal((t, feat)) * 3).astype(np.float32) for t in (2, 130, 25, 257, 1)]
    big = small[:4] + [(rng.standard_normal((300, feat)) * 3).astype(np.float32) for _ in range(85)] + [small[4]]
    outs_s, outs_b = run(small), run(big)
    for m, a_, c_ in zip(small, outs_s, outs_b[:4] + outs_b[-1:]):
        assert oracle.rel_l2(a_, oracle.tdnn_layer(m, w, b, bn, act, alpha, 1, np.float64)) < TOL_GEMM
        assert np.array_equal(a_, c_)
    for i in (4, 40, 88):
        assert oracle.rel_l2(outs_b[i], oracle.tdnn_layer(big[i], w, b, bn, act, alpha, 1, np.float64)) < TOL_GEMM


# ---- the whole network in the "fp32tc" arithmetic (engine.DeviceModel(precision="fp32tc")) -----------------------------------
TOL_XVEC = 2e-6           # x-vector, relative L2 against the fp64 oracle (measured ~3e-7; the exact-fp32 kernels ~2e-7)


@pytest.fixture(scope="module")
def net(oracle_mod):
    import torch
    from xvector_amd import engine, hiplib, synthetic, topology
    hiplib.require_gpu()
    return dict(torch=torch, hiplib=hiplib, engine=engine, oracle=oracle_mod, synthetic=synthetic, topology=topology)


def test_fp32tc_forward_matches_golden(net, golden):
    """The golden x-vectors of the default topology (tests/golden/forward_refgraph.npz: what the reference's own graph returned under
    tests/golden/numpy_tf1.py, on seeded weights / inputs)."""
    g = golden("forward_refgraph.npz")
    seed = int(g["seed"])
    topo = net["topology"].get("ModelWithoutDropout")
    w = net["synthetic"].trained_like(topo, 23, seed=seed)
    rng = np.random.default_rng(seed + 1)
    Ts = [25, 200, 400, 1000]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in Ts]
    for emb_idx in (0, 1):
        model = net["engine"].DeviceModel(w, topo, "cuda:0", embedding_index=emb_idx, precision="fp32tc")
        assert model.toom and model.arithmetic == "fp32tc" and model.align == 8
        assert [isinstance(L["wp"], net["hiplib"].PackedToom) for L in model.layers] == [False, True, True, False, False]
        assert isinstance(model.layers[0]["wp"], net["hiplib"].PackedRows)
        vecs = net["engine"].Extractor(model, 1, -1).extract(mats)
        for T, v in zip(Ts, vecs):
            ref32 = net["oracle"].chunk_average(g["default_T%d_e%d" % (T, emb_idx)].astype(np.float32)[None, :], [T], np.float32)
            assert net["oracle"].rel_l2(v, ref32) < TOL_XVEC, (T, emb_idx)


def test_fp32tc_ragged_batches_chunking_and_bits(net, default_weights):
    """Config-3 style lengths with chunking on and a small batch budget (several batches) against the fp64 oracle; and the
    size-independent property: an utterance's x-vector is bit-identical alone, in a batch, in any order."""
    topo, w = default_weights
    oracle = net["oracle"]
    rng = np.random.default_rng(78)
    lens = [25, 26, 31, 100, 257, 999, 1024, 2300, 24, 0, 613]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in lens]
    model = net["engine"].DeviceModel(w, topo, "cuda:0", precision="fp32tc")
    ex = net["engine"].Extractor(model, 25, 1000, max_batch_rows=1500)
    vecs = ex.extract(mats)
    assert ex.stats["batches"] > 2
    for T, m, v in zip(lens, mats, vecs):
        ref = oracle.embed_utterance(m, w, topo, 25, 1000, np.float64)
        if T < 25:
            assert v is None and ref is None
            continue
        assert oracle.rel_l2(v, ref) < TOL_XVEC, T
    lens2 = rng.integers(200, 401, size=40)
    mats2 = [(rng.standard_normal((int(T), 23)) * 3.0).astype(np.float32) for T in lens2]
    ex2 = net["engine"].Extractor(model, 25, 10000)
    together = ex2.extract(mats2)
    perm = rng.permutation(len(mats2))
    shuffled = ex2.extract([mats2[i] for i in perm])
    for j, i in enumerate(perm):
        assert np.array_equal(together[i], shuffled[j])
    for i in (0, 7, 39):
        assert np.array_equal(ex2.extract([mats2[i]])[0], together[i])


def test_fp32tc_dilated_topology_runs_f23_on_strided_rows(net, golden):
    """The dilated class (models.py:538-639: kernels [5,3,3,1,1], dilations [1,2,3,1,1]): its K = 3 layers run as F(2, 3) over every
    2nd / 3rd row (xv_tdnn_layer_toom_dilated_f32), batches are laid out on multiples of lcm(8, 4, 6) = 24 rows, and the x-vectors
    agree with what the reference's own graph returned (forward_refgraph.npz) and with the fp64 oracle on ragged chunked batches;
    an utterance's bits do not depend on its batch."""
    g = golden("forward_refgraph.npz")
    seed = int(g["seed"])
    topo = net["topology"].get("ModelWithoutDropoutTdnn")
    w = net["synthetic"].trained_like(topo, 23, seed=seed)
    model = net["engine"].DeviceModel(w, topo, "cuda:0", precision="fp32tc")
    assert model.toom and model.align == 24
    assert [type(L["wp"]).__name__ for L in model.layers][:3] == ["PackedRows", "PackedToom", "PackedToom"]
    rng = np.random.default_rng(seed + 1)
    Ts = [25, 200, 400, 1000]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in Ts]
    vecs = net["engine"].Extractor(model, 1, -1).extract(mats)
    for T, v in zip(Ts, vecs):
        ref32 = net["oracle"].chunk_average(g["dilated_T%d_e0" % T].astype(np.float32)[None, :], [T], np.float32)
        assert net["oracle"].rel_l2(v, ref32) < TOL_XVEC, T
    lens = [25, 26, 31, 100, 257, 999, 1024, 2300, 24, 613]
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in lens]
    ex = net["engine"].Extractor(model, 25, 1000, max_batch_rows=1500)
    vecs = ex.extract(mats)
    direct = net["engine"].Extractor(net["engine"].DeviceModel(w, topo, "cuda:0", precision="fp32"), 25, 1000).extract(mats)
    for T, m, v, d in zip(lens, mats, vecs, direct):
        ref = net["oracle"].embed_utterance(m, w, topo, 25, 1000, np.float64)
        if T < 25:
            assert v is None and ref is None
            continue
        assert net["oracle"].rel_l2(v, ref) < TOL_XVEC, T
        assert not np.array_equal(v, d)                     # (another rounding than the direct kernels': the Toom-Cook form did run)
    ex2 = net["engine"].Extractor(model, 25, 10000)
    together = ex2.extract(mats)
    perm = rng.permutation(len(mats))
    shuffled = ex2.extract([mats[i] for i in perm])
    for j, i in enumerate(perm):
        assert (together[i] is None and shuffled[j] is None) or np.array_equal(together[i], shuffled[j])
    assert np.array_equal(ex2.extract([mats[4]])[0], together[4])


def test_fp32tc_other_topologies_take_what_the_kernel_covers(net):
    """A topology none of whose layers the Toom-Cook kernel takes (48-channel layers: no whole 32-channel slabs; layer 0 has 23
    input channels and takes the rows form, which keeps the direct kernel's products and order): fp32tc then IS the exact-fp32
    path, bit for bit."""
    topo = net["topology"].get("ModelWithoutDropoutTdnn")
    topo["layer_sizes"] = [48, 48, 48, 48, 96]
    w = net["synthetic"].trained_like(topo, 23, seed=5)
    rng = np.random.default_rng(6)
    mats = [(rng.standard_normal((T, 23)) * 3.0).astype(np.float32) for T in (40, 211)]
    m = net["engine"].DeviceModel(w, topo, "cuda:0", precision="fp32tc")
    assert not any(isinstance(L["wp"], net["hiplib"].PackedToom) for L in m.layers) and m.align == 8
    a = net["engine"].Extractor(m, 25, 10000).extract(mats)
    b = net["engine"].Extractor(net["engine"].DeviceModel(w, topo, "cuda:0", precision="fp32"), 25, 10000).extract(mats)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
