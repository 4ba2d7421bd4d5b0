"""GPU parity tests, kernel by kernel, through the C ABI (ctypes) against the CPU oracle.

Tolerance: the north star asks for 1e-4 relative L2 on the final embedding; the per-kernel checks here
are much tighter (fp32 MFMA is an exact-product fp32 fma chain): 2e-6 relative L2 against the fp64
oracle for the GEMM kernels, 2e-6 for pooling, bit-exact for the chunk average.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_GEMM = 2e-6          # exact-fp32 MFMA path
TOL_GEMM3 = 2e-5         # bf16x3 split path (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-17 per operand
TOL_POOL = 2e-6


@pytest.fixture(scope="module")
def env(oracle_mod):
    import torch
    from xvector_amd import engine, hiplib
    hiplib.require_gpu()
    return dict(torch=torch, hiplib=hiplib, engine=engine, oracle=oracle_mod, dev=torch.device("cuda:0"))


def _rand_bn(rng, c):
    return ((1 + 0.1 * rng.standard_normal(c)).astype(np.float32), (0.1 * rng.standard_normal(c)).astype(np.float32),
            (0.2 * rng.standard_normal(c)).astype(np.float32), np.exp(0.2 * rng.standard_normal(c)).astype(np.float32))


def _run_layer(env, mats, w, b, bn, act, alpha, K, dil, preact=False, precision="fp32", fmt="f32"):
    """Pack `mats` with gap rows, run xv_tdnn_layer_f32 / _bf16x3 (fmt: tensor format of x and y on the bf16x3
    path), return per-chunk outputs and the full y as fp32."""
    torch, hiplib, engine, dev = env["torch"], env["hiplib"], env["engine"], env["dev"]
    gap = max(1, (K - 1) * dil // 2)
    layout = engine.BatchLayout([m.shape[0] for m in mats], gap)
    host = np.zeros((layout.rows, mats[0].shape[1]), np.float32)
    layout.pack(mats, host)
    x = torch.from_numpy(host).to(dev)
    rv = torch.from_numpy(layout.row_valid()).to(dev)
    cin, cout = w.shape[1], w.shape[2]
    if precision == "bf16x3":
        wp = hiplib.pack_weights_bf16x3(torch.from_numpy(np.ascontiguousarray(w)).to(dev))
    else:
        wp = hiplib.pack_weights(torch.from_numpy(np.ascontiguousarray(w.reshape(K * cin, cout))).to(dev))
    bias = torch.from_numpy(b).to(dev)
    scale = shift = None
    if bn is not None:
        scale, shift = hiplib.fold_bn(*(torch.from_numpy(a).to(dev) for a in bn), 1e-3)
    al = None if alpha is None else torch.from_numpy(np.atleast_1d(alpha).astype(np.float32)).to(dev)
    # poison the output so that unwritten elements are caught
    y = torch.full((layout.rows, cout), float("nan"), dtype=torch.float32, device=dev)
    ypre = torch.full_like(y, float("nan")) if preact else None
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]
    if precision == "bf16x3" and fmt == "split":
        xin = hiplib.SplitBuf(layout.rows, cin, dev)
        hiplib.split_encode(x, xin)
        yout = hiplib.SplitBuf(layout.rows, cout, dev)
        hiplib.tdnn_layer(xin, wp, bias, scale, shift, code, al, K, dil, rv, yout, ypre, rows=layout.rows)
        y = hiplib.split_decode(yout, layout.rows)
    else:
        hiplib.tdnn_layer(x, wp, bias, scale, shift, code, al, K, dil, rv, y, ypre)
    torch.cuda.synchronize()
    yh = y.cpu().numpy()
    outs = [yh[s:s + n] for s, n in zip(layout.row_start, layout.row_len)]
    return outs, yh, layout, (ypre.cpu().numpy() if preact else None)


@pytest.mark.parametrize("cin,cout,K,dil,act", [
    (23, 512, 5, 1, "relu"),        # layer 0: scalar-load path, Cin not a multiple of 4
    (512, 512, 5, 1, "relu"),       # layer 1
    (512, 512, 7, 1, "relu"),       # layer 2
    (512, 512, 1, 1, "relu"),       # layer 3
    (512, 1536, 1, 1, "relu"),      # layer 4
    (512, 512, 3, 2, "relu"),       # dilated variant layer 1
    (512, 512, 3, 3, "relu"),       # dilated variant layer 2
    (64, 48, 5, 1, "prelu"),        # Cout not a multiple of 128/32, PReLU epilogue
    (40, 200, 3, 1, "lrelu"),       # Cin not a multiple of 32 (vector path with a ragged last slab)
    (5, 32, 5, 1, "none"),
])
def test_tdnn_layer_matches_oracle(env, cin, cout, K, dil, act):
    oracle = env["oracle"]
    rng = np.random.default_rng(cin * 1000 + cout + K * 7 + dil)
    lens = [25, 1, 130, 257, 64, 3]           # chunks shorter than the halo and spanning tile boundaries
    mats = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in lens]
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    alpha = None
    if act == "lrelu":
        alpha = np.array([0.2], np.float32)
    elif act == "prelu":
        alpha = (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
    outs, yh, layout, _ = _run_layer(env, mats, w, b, bn, act, alpha, K, dil)
    for m, got in zip(mats, outs):
        ref = oracle.tdnn_layer(m, w, b, bn, act, alpha, dil, np.float64)
        assert np.isfinite(got).all()
        assert oracle.rel_l2(got, ref) < TOL_GEMM
    # gap rows must be written as exact zeros (the invariant the next layer relies on)
    valid = layout.row_valid().astype(bool)
    assert (yh[~valid] == 0).all()


def test_fp32_gemm_both_tile_heights(env):
    """The exact-fp32 GEMM picks 64-row workgroup tiles when 128-row tiles would not fill 1.5 rounds of the chip and 128-row
    tiles otherwise: the same layer on a small batch (64-row form) and on a 26 k-row batch (128-row form) against the oracle,
    and the chunks common to both batches must come out bit-identical (tile height must not change the arithmetic)."""
    oracle = env["oracle"]
    rng = np.random.default_rng(9)
    cin, cout, K = 64, 512, 5
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    small = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in (130, 25, 257)]
    big = small + [(rng.standard_normal((300, cin)) * 2).astype(np.float32) for _ in range(85)]
    outs_s, _, lay_s, _ = _run_layer(env, small, w, b, bn, "relu", None, K, 1)
    outs_b, _, lay_b, _ = _run_layer(env, big, w, b, bn, "relu", None, K, 1)
    assert (lay_s.rows + 127) // 128 * 4 < 768 <= (lay_b.rows + 127) // 128 * 4
    for m, a, c in zip(small, outs_s, outs_b):
        assert oracle.rel_l2(a, oracle.tdnn_layer(m, w, b, bn, "relu", None, 1, np.float64)) < TOL_GEMM
        assert np.array_equal(a, c)
    for i in (3, 40, 87):
        assert oracle.rel_l2(outs_b[i], oracle.tdnn_layer(big[i], w, b, bn, "relu", None, 1, np.float64)) < TOL_GEMM


@pytest.mark.parametrize("fmt", ["f32", "split"])
@pytest.mark.parametrize("cin,cout,K,dil,act", [
    (24, 512, 5, 1, "relu"),        # layer 0 with the 23 MFCC dims padded to 24 columns
    (512, 512, 5, 1, "relu"),
    (512, 512, 7, 1, "relu"),
    (512, 1536, 1, 1, "relu"),
    (512, 512, 3, 3, "relu"),       # dilated
    (64, 48, 5, 1, "prelu"),        # ragged Cout
    (40, 200, 3, 2, "lrelu"),       # Cin not a multiple of 32
    (96, 64, 5, 1, "relu"),         # three 32-channel slabs: the 16 x 16 MFMA form needs an even number, this one runs on 32 x 32 tiles
])
def test_tdnn_layer_bf16x3_matches_oracle(env, cin, cout, K, dil, act, fmt):
    oracle = env["oracle"]
    rng = np.random.default_rng(cin * 1000 + cout + K * 7 + dil + 1)
    lens = [25, 1, 130, 257, 64, 3]
    mats = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in lens]
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    alpha = None
    if act == "lrelu":
        alpha = np.array([0.2], np.float32)
    elif act == "prelu":
        alpha = (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
    outs, yh, layout, _ = _run_layer(env, mats, w, b, bn, act, alpha, K, dil, precision="bf16x3", fmt=fmt)
    for m, got in zip(mats, outs):
        ref = oracle.tdnn_layer(m, w, b, bn, act, alpha, dil, np.float64)
        assert np.isfinite(got).all()
        assert oracle.rel_l2(got, ref) < TOL_GEMM3
    assert (yh[~layout.row_valid().astype(bool)] == 0).all()
    # batch-1 == batched, bit for bit, also on the split path
    alone, _, _, _ = _run_layer(env, [mats[2]], w, b, bn, act, alpha, K, dil, precision="bf16x3", fmt=fmt)
    assert np.array_equal(alone[0], outs[2])


@pytest.mark.parametrize("cin,cout,K,dil,act", [
    (512, 512, 1, 1, "relu"),
    (512, 1536, 1, 1, "relu"),
    (512, 512, 3, 3, "relu"),
    (512, 512, 5, 1, "prelu"),
    (512, 512, 7, 1, "relu"),
    (64, 200, 3, 1, "lrelu"),       # ragged Cout
])
def test_bf16x3_256_row_tiles_equal_128_row_tiles_bitwise(env, cin, cout, K, dil, act):
    """XV_TUNE_TILE_ROWS: the 8-wave / 256-row workgroup tile accumulates every output element in the same order as the
    4-wave / 128-row tile, so layer outputs and pooling block statistics must be bit-identical (and the 128-row form is
    the one checked against the oracle above).  Row counts straddle tile boundaries (R % 256 in 1..255)."""
    torch, hiplib, engine, dev = env["torch"], env["hiplib"], env["engine"], env["dev"]
    rng = np.random.default_rng(cin + cout * 3 + K)
    lens = [25, 1, 130, 257, 64, 3, 300, 511, 77]
    mats = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in lens]
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    alpha = None
    if act == "lrelu":
        alpha = np.array([0.2], np.float32)
    elif act == "prelu":
        alpha = (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
    layout = engine.BatchLayout(lens, max(1, (K - 1) * dil // 2), hiplib.POOL_BLOCK_ROWS)
    host = np.zeros((layout.rows, cin), np.float32)
    layout.pack(mats, host)
    x = torch.from_numpy(host).to(dev)
    rv = torch.from_numpy(layout.row_valid()).to(dev)
    wp = hiplib.pack_weights_bf16x3(torch.from_numpy(w).to(dev))
    scale, shift = hiplib.fold_bn(*(torch.from_numpy(a).to(dev) for a in bn), 1e-3)
    al = None if alpha is None else torch.from_numpy(alpha).to(dev)
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]
    bias = torch.from_numpy(b).to(dev)
    xin = hiplib.SplitBuf(layout.rows, cin, dev)
    hiplib.split_encode(x, xin)
    got = {}
    try:
        for rows in (128, 256):
            hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, rows)
            yout = hiplib.SplitBuf(layout.rows, cout, dev)
            hiplib.tdnn_layer(xin, wp, bias, scale, shift, code, al, K, dil, rv, yout, None, rows=layout.rows)
            blk = torch.full((hiplib.block_stats_floats(layout.rows, cout),), float("nan"), dtype=torch.float32, device=dev)
            hiplib.tdnn_layer_pool(xin, layout.rows, wp, bias, scale, shift, code, al, dil, rv, blk)
            torch.cuda.synchronize()
            got[rows] = (hiplib.split_decode(yout, layout.rows).cpu().numpy(), blk.cpu().numpy())
    finally:
        hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, 0)
    assert np.isfinite(got[256][0]).all() and np.abs(got[256][0]).max() > 0
    assert np.array_equal(got[128][0], got[256][0])
    assert np.array_equal(got[128][1], got[256][1], equal_nan=True)


def test_pack_weights_bf16x3_many_equals_the_single_packs(env):
    """xv_pack_weights_bf16x3_many: every layer's forward tiles == xv_pack_weights_bf16x3(w padded to cin_pad), its input-gradient
    tiles == xv_pack_weights_bf16x3(w'[k, o, c] = w[K-1-k, c, o]) -- byte for byte; and a second call follows the weights."""
    torch, hiplib, dev = env["torch"], env["hiplib"], env["dev"]
    g = torch.Generator().manual_seed(7)
    shapes = [(5, 23, 24, 64), (5, 64, 64, 64), (7, 64, 64, 96), (1, 96, 96, 200), (1, 200, 200, 32), (1, 32, 32, 10), (3, 40, 40, 512)]
    ws = [torch.randn((K, cin, cout), generator=g).to(dev) for K, cin, _, cout in shapes]
    plan = hiplib.PackPlan([(w, pad, i != 5) for i, (w, (_, _, pad, _)) in enumerate(zip(ws, shapes))])
    for rnd in range(2):
        plan.repack()
        for i, (w, (K, cin, pad, cout)) in enumerate(zip(ws, shapes)):
            wpad = torch.cat([w, torch.zeros((K, pad - cin, cout), device=dev)], dim=1) if pad != cin else w
            ref_f = hiplib.pack_weights_bf16x3(wpad.contiguous())
            assert (plan.fwd[i].K, plan.fwd[i].cin, plan.fwd[i].cout) == (K, pad, cout)
            assert torch.equal(plan.fwd[i].wt, ref_f.wt), (rnd, i)
            if i == 5:
                assert plan.bwd[i] is None
                continue
            ref_b = hiplib.pack_weights_bf16x3(wpad.flip(0).permute(0, 2, 1).contiguous())
            assert (plan.bwd[i].K, plan.bwd[i].cin, plan.bwd[i].cout) == (K, cout, pad)
            assert torch.equal(plan.bwd[i].wt, ref_b.wt), (rnd, i)
        for w in ws:
            w.mul_(1.5).add_(0.25)               # (in place: the plan holds the pointers, as the trainer's flat buffer)


def test_split_format_roundtrip_and_layout(env):
    """xv_split_encode_f32 / xv_split_decode_f32 against a NumPy statement of the documented layout
    (include/xvector_hip.h): slot t = plane*4 + (k>>3) stored at t ^ ((r>>1)&7), value = hi + lo."""
    torch, hiplib, dev = env["torch"], env["hiplib"], env["dev"]
    rng = np.random.default_rng(21)
    R, C = 37, 72                                   # 3 slabs, last one ragged
    x = (rng.standard_normal((R, C)) * 5).astype(np.float32)
    buf = hiplib.SplitBuf(R, C, dev)
    hiplib.split_encode(torch.from_numpy(x).to(dev), buf)
    back = hiplib.split_decode(buf, R).cpu().numpy()
    assert np.abs(back - x).max() <= np.abs(x).max() * 2.0 ** -16
    raw = buf.base.cpu().numpy()[hiplib.SPLIT_PAD_BEFORE * buf.row_bytes:].view(np.uint16)

    def bf16_rne(a):
        u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)

    hi = bf16_rne(x)
    lo = bf16_rne(x - (hi.astype(np.uint32) << 16).view(np.float32))
    for r in (0, 1, 2, 9, 36):
        for c in (0, 7, 8, 31, 32, 71):
            s, k = c >> 5, c & 31
            for plane, want in ((0, hi[r, c]), (1, lo[r, c])):
                phys = (plane * 4 + (k >> 3)) ^ ((r >> 1) & 7)
                got = raw[((r * 3 + s) * 128 + phys * 16) // 2 + (k & 7)]
                assert got == want, (r, c, plane)
    assert not buf.base[:hiplib.SPLIT_PAD_BEFORE * buf.row_bytes].any()          # padding rows stay zero


def test_bf16x3_rejects_unaligned_fp32_input(env):
    torch, hiplib, dev = env["torch"], env["hiplib"], env["dev"]
    x = torch.zeros((10, 23), dtype=torch.float32, device=dev)
    wp = hiplib.pack_weights_bf16x3(torch.zeros((1, 23, 8), dtype=torch.float32, device=dev))
    y = torch.zeros((10, 8), dtype=torch.float32, device=dev)
    with pytest.raises(hiplib.XvectorHipError):
        hiplib.tdnn_layer(x, wp, None, None, None, 1, None, 1, 1, None, y)       # fp32 rows need Cin % 4 == 0


def test_tdnn_layer_batch1_equals_batched_bitwise(env):
    """The reference runs batch 1 (models.py:410-414); a chunk's rows must not depend on its batch
    neighbours: identical bits alone and inside a batch."""
    rng = np.random.default_rng(5)
    K, cin, cout = 5, 512, 512
    mats = [(rng.standard_normal((t, cin))).astype(np.float32) for t in (200, 37, 311)]
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    outs, _, _, _ = _run_layer(env, mats, w, b, bn, "relu", None, K, 1)
    for m, got in zip(mats, outs):
        alone, _, _, _ = _run_layer(env, [m], w, b, bn, "relu", None, K, 1)
        assert np.array_equal(alone[0], got)


def test_tdnn_preact_output_and_no_bn(env):
    oracle = env["oracle"]
    rng = np.random.default_rng(11)
    mats = [(rng.standard_normal((90, 128))).astype(np.float32)]
    w = (rng.standard_normal((1, 128, 256)) / np.sqrt(128)).astype(np.float32)
    b = (0.1 * rng.standard_normal(256)).astype(np.float32)
    outs, yh, layout, pre = _run_layer(env, mats, w, b, None, "relu", None, 1, 1, preact=True)
    ref_pre = oracle.tdnn_layer(mats[0], w, b, None, "none", None, 1, np.float64)
    s = layout.row_start[0]
    assert oracle.rel_l2(pre[s:s + 90], ref_pre) < TOL_GEMM
    assert oracle.rel_l2(outs[0], np.maximum(ref_pre, 0)) < TOL_GEMM


@pytest.mark.parametrize("C,lens,split", [
    (1536, [25, 200, 400, 1, 7, 33, 512], 512),       # direct path (max_len <= split)
    (1536, [25, 2000, 513, 10000, 100], 512),         # split path + merge kernel
    (48, [5, 64, 300], 512),                          # C < 64: partially filled wave
    (1536, [1000, 999], 128),
])
def test_stats_pool_matches_oracle(env, C, lens, split):
    torch, hiplib, engine, oracle, dev = env["torch"], env["hiplib"], env["engine"], env["oracle"], env["dev"]
    rng = np.random.default_rng(C + sum(lens))
    # post-ReLU/BN-like data: point mass + large per-channel mean offsets (stresses the variance)
    mats = [(np.maximum(rng.standard_normal((t, C)), 0) * 1.7 + 3.0 * rng.standard_normal(C)).astype(np.float32) for t in lens]
    layout = engine.BatchLayout(lens, 3)
    host = np.full((layout.rows, C), 1e6, np.float32)      # garbage in the gaps must not matter
    for s, m in zip(layout.row_start, mats):
        host[s:s + m.shape[0]] = m
    h = torch.from_numpy(host).to(dev)
    rs = torch.from_numpy(layout.row_start).to(dev)
    rl = torch.from_numpy(layout.row_len).to(dev)
    out = torch.full((len(lens), 2 * C), float("nan"), dtype=torch.float32, device=dev)
    need = hiplib.stats_pool_workspace_bytes(C, len(lens), max(lens), split)
    assert (need > 0) == (max(lens) > split)
    ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=dev)
    hiplib.stats_pool(h, rs, rl, len(lens), max(lens), split, 1e-5, out, ws)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i, m in enumerate(mats):
        ref = oracle.stats_pool(m, 1e-5, np.float64)
        assert oracle.rel_l2(got[i, :C], ref[:C]) < TOL_POOL
        assert oracle.rel_l2(got[i, C:], ref[C:]) < TOL_POOL


@pytest.mark.parametrize("cin,cout,K,dil,act,lens", [
    (512, 1536, 1, 1, "relu", [25, 1, 7, 8, 9, 130, 257, 1000]),      # layer 4 of the default topology (64-row tiles: a small batch)
    (512, 1536, 1, 1, "relu", [300] * 90 + [25, 1, 7]),                # ... on 128-row tiles
    (64, 200, 3, 1, "prelu", [300, 25, 64]),                          # ragged last column tile, K > 1
    (40, 48, 5, 2, "lrelu", [1200, 33]),
])
def test_tdnn_layer_pool_f32_blocks_match_oracle(env, cin, cout, K, dil, act, lens):
    """xv_tdnn_layer_pool_f32 (the exact-fp32 GEMM with the block-statistics epilogue) + xv_stats_pool_blocks_f32 == statistics
    pooling of the layer output in the fp64 oracle, at the tolerance of the fp32 GEMM; the block statistics of a chunk do not
    depend on where it sits in the batch (a second batch with the chunks in reverse order: bit-identical pooled rows)."""
    torch, hiplib, engine, oracle, dev = env["torch"], env["hiplib"], env["engine"], env["oracle"], env["dev"]
    rng = np.random.default_rng(cin + cout + K + len(lens))
    mats = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in lens]
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    alpha = None
    if act == "lrelu":
        alpha = np.array([0.2], np.float32)
    elif act == "prelu":
        alpha = (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
    gap = max(1, (K - 1) * dil // 2)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    wp = hiplib.pack_weights(t(w.reshape(K * cin, cout)))
    scale, shift = hiplib.fold_bn(*(t(a) for a in bn), 1e-3)
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]

    def run(ms):
        layout = engine.BatchLayout([m.shape[0] for m in ms], gap, hiplib.POOL_BLOCK_ROWS)
        host = np.zeros((layout.rows, cin), np.float32)
        layout.pack(ms, host)
        blk = torch.full((hiplib.block_stats_floats(layout.rows, cout),), float("nan"), dtype=torch.float32, device=dev)
        hiplib.tdnn_layer_pool(t(host), layout.rows, wp, t(b), scale, shift, code, t(alpha), dil, t(layout.row_valid()), blk, K=K)
        out = torch.full((len(ms), 2 * cout), float("nan"), dtype=torch.float32, device=dev)
        hiplib.stats_pool_blocks(blk, cout, t(layout.row_start), t(layout.row_len), len(ms), 1e-5, out)
        torch.cuda.synchronize()
        return out.cpu().numpy()
    got = run(mats)
    assert np.isfinite(got).all()
    for i in list(range(min(len(mats), 8))) + [len(mats) - 1]:
        ref = oracle.stats_pool(oracle.tdnn_layer(mats[i], w, b, bn, act, alpha, dil, np.float64), 1e-5, np.float64)
        assert oracle.rel_l2(got[i, :cout], ref[:cout]) < TOL_GEMM, (i, lens[i])
        assert oracle.rel_l2(got[i, cout:], ref[cout:]) < TOL_GEMM, (i, lens[i])
    if len(mats) < 20:                                             # (same tile height in both runs)
        assert np.array_equal(run(mats[::-1])[::-1], got)


@pytest.mark.parametrize("fmt", ["f32", "split"])
@pytest.mark.parametrize("cin,cout,K,dil,act,lens", [
    (512, 1536, 1, 1, "relu", [25, 1, 7, 8, 9, 130, 257, 1000]),      # layer 4 of the default topology
    (64, 200, 3, 1, "prelu", [300, 25, 64]),                          # ragged Cout, K > 1
    (40, 48, 5, 2, "lrelu", [1200, 33]),                               # long chunk: many blocks per chunk
])
def test_tdnn_layer_pool_blocks_match_oracle(env, fmt, cin, cout, K, dil, act, lens):
    """xv_tdnn_layer_pool_bf16x3 + xv_stats_pool_blocks_f32 == statistics pooling of the layer output (fp64 oracle)."""
    torch, hiplib, engine, oracle, dev = env["torch"], env["hiplib"], env["engine"], env["oracle"], env["dev"]
    rng = np.random.default_rng(cin + cout + K + len(lens))
    mats = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in lens]
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    alpha = None
    if act == "lrelu":
        alpha = np.array([0.2], np.float32)
    elif act == "prelu":
        alpha = (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
    gap = max(1, (K - 1) * dil // 2)
    layout = engine.BatchLayout(lens, gap, hiplib.POOL_BLOCK_ROWS)
    assert (layout.row_start % 8 == 0).all()
    host = np.zeros((layout.rows, cin), np.float32)
    layout.pack(mats, host)
    x = torch.from_numpy(host).to(dev)
    rv = torch.from_numpy(layout.row_valid()).to(dev)
    wp = hiplib.pack_weights_bf16x3(torch.from_numpy(w).to(dev))
    scale, shift = hiplib.fold_bn(*(torch.from_numpy(a).to(dev) for a in bn), 1e-3)
    al = None if alpha is None else torch.from_numpy(alpha).to(dev)
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]
    xin = x
    if fmt == "split":
        xin = hiplib.SplitBuf(layout.rows, cin, dev)
        hiplib.split_encode(x, xin)
    blk = torch.full((hiplib.block_stats_floats(layout.rows, cout),), float("nan"), dtype=torch.float32, device=dev)
    hiplib.tdnn_layer_pool(xin, layout.rows, wp, torch.from_numpy(b).to(dev), scale, shift, code, al, dil, rv, blk)
    out = torch.full((len(lens), 2 * cout), float("nan"), dtype=torch.float32, device=dev)
    rs, rl = torch.from_numpy(layout.row_start).to(dev), torch.from_numpy(layout.row_len).to(dev)
    hiplib.stats_pool_blocks(blk, cout, rs, rl, len(lens), 1e-5, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    for i, m in enumerate(mats):
        ref = oracle.stats_pool(oracle.tdnn_layer(m, w, b, bn, act, alpha, dil, np.float64), 1e-5, np.float64)
        assert oracle.rel_l2(got[i, :cout], ref[:cout]) < TOL_GEMM3, (i, lens[i])
        assert oracle.rel_l2(got[i, cout:], ref[cout:]) < TOL_GEMM3, (i, lens[i])
    # a chunk that does not start on a block boundary is refused loudly (NaN), its neighbours are unaffected
    rs_bad = layout.row_start.copy()
    rs_bad[1] += 1
    hiplib.stats_pool_blocks(blk, cout, torch.from_numpy(rs_bad).to(dev), rl, len(lens), 1e-5, out)
    bad = out.cpu().numpy()
    assert np.isnan(bad[1]).all() and np.array_equal(bad[0], got[0]) and np.array_equal(bad[2:], got[2:])


@pytest.mark.parametrize("feat,cout,K,dil,act,lens", [
    (23, 512, 5, 1, "relu", [25, 1, 7, 130, 257, 600, 64, 3]),      # layer 0 of the default topology (23 MFCCs in 24 columns)
    (5, 64, 3, 2, "prelu", [300, 25, 2]),                            # narrow features, dilation, one pass of 64 channels
    (30, 288, 3, 1, "lrelu", [1200, 33]),                            # two passes (256 + 32 channels), 3 taps x 32 columns
    (23, 512, 5, 1, "none", [513]),
])
def test_tdnn_first_layer_kernel_matches_oracle_and_the_general_kernel(env, feat, cout, K, dil, act, lens):
    """xv_tdnn_first_bf16x3 (im2col operand built in registers from the fp32 rows, split-format output straight from the
    accumulators) against the fp64 oracle, and against xv_tdnn_layer_bf16x3 on the same input: same products, another
    fp32 summation order.  Gap rows come out as exact zeros; rows past R are not written."""
    torch, hiplib, engine, oracle, dev = env["torch"], env["hiplib"], env["engine"], env["oracle"], env["dev"]
    in_dim = (feat + 7) // 8 * 8
    assert hiplib.first_supported(K, in_dim, cout) and not hiplib.first_supported(7, 24, 512) and not hiplib.first_supported(5, 24, 520)
    rng = np.random.default_rng(feat + cout + K)
    mats = [(rng.standard_normal((t, feat)) * 3).astype(np.float32) for t in lens]
    w = (rng.standard_normal((K, feat, cout)) / np.sqrt(K * feat)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn = _rand_bn(rng, cout)
    alpha = None
    if act == "lrelu":
        alpha = np.array([0.2], np.float32)
    elif act == "prelu":
        alpha = (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
    gap = max(1, (K - 1) * dil // 2)
    layout = engine.BatchLayout(lens, gap, 8)
    host = np.zeros((layout.rows, in_dim), np.float32)
    layout.pack(mats, host[:, :feat] if False else host)           # pack writes the first `feat` columns of each row
    x = torch.from_numpy(host).to(dev)
    rv = torch.from_numpy(layout.row_valid()).to(dev)
    wpad = np.zeros((K, in_dim, cout), np.float32)
    wpad[:, :feat] = w
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    scale, shift = hiplib.fold_bn(*(t(a) for a in bn), 1e-3)
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]
    first = hiplib.pack_first_bf16x3(t(wpad))
    y = hiplib.SplitBuf(layout.rows, cout, dev)
    y.base.fill_(0x7f)                                             # poison (bf16 0x7f7f = 3.4e38): unwritten slots would show
    hiplib.tdnn_first(x, layout.rows, first, t(b), scale, shift, code, t(alpha), dil, rv, y)
    got = hiplib.split_decode(y, layout.rows).cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got).max() < 1e6
    valid = layout.row_valid().astype(bool)
    assert (got[~valid] == 0).all()
    for i, m in enumerate(mats):
        ref = oracle.tdnn_layer(m, w, b, bn, act, alpha, dil, np.float64)
        s = int(layout.row_start[i])
        assert oracle.rel_l2(got[s:s + lens[i]], ref) < TOL_GEMM3, (i, lens[i])
    y2 = hiplib.SplitBuf(layout.rows, cout, dev)
    hiplib.tdnn_layer(x, hiplib.pack_weights_bf16x3(t(wpad)), t(b), scale, shift, code, t(alpha), K, dil, rv, y2, None, rows=layout.rows)
    two = hiplib.split_decode(y2, layout.rows).cpu().numpy()
    assert oracle.rel_l2(got, two) < 5e-6
    # rows past R stay untouched: run on a prefix of the rows and look at the poison behind it
    R2 = int(layout.row_start[-1])                                 # everything before the last chunk
    y3 = hiplib.SplitBuf(layout.rows, cout, dev)
    y3.base.fill_(0x7f)
    hiplib.tdnn_first(x, R2, first, t(b), scale, shift, code, t(alpha), dil, rv, y3)
    tail = hiplib.split_decode(y3, layout.rows).cpu().numpy()[R2:]
    assert (np.abs(tail) > 1e30).all()


@pytest.mark.parametrize("fmt", ["split8", "split"])
@pytest.mark.parametrize("cout,tiles", [(512, 1), (512, 2), (512, 3), (512, 5), (512, 8), (288, 7), (512, 0)])
def test_tdnn_first_layer_any_number_of_tiles_per_wave(env, cout, tiles, fmt):
    """The first-layer kernel gives every wave a run of 16-frame tiles and walks through them three at a time while the weights of
    the four 128-channel passes alternate between two LDS buffers, fetched a step ahead and awaited with a COUNTED vmcnt when the
    wave issued a full step's stores behind the fetch (24 or 16).  XV_TUNE_FIRST_TILES forces the run length: 1 and 2 (one group, the
    drain path), 3 (vmcnt(24) between the passes), 5 (groups of 3 + 2: both counted waits), 7 / 8 (three groups, Cout = 288: a
    ragged last pass), 0 = the launcher's own choice -- all against the general bf16x3 kernel on 6 k rows that end in the middle of
    a tile, gap rows included, bit for bit between the run lengths."""
    torch, hiplib, engine, oracle, dev = env["torch"], env["hiplib"], env["engine"], env["oracle"], env["dev"]
    K, feat, in_dim, dil = 5, 23, 24, 1
    rng = np.random.default_rng(cout)
    lens = [int(v) for v in rng.integers(1, 700, 18)]
    mats = [(rng.standard_normal((t, feat)) * 3).astype(np.float32) for t in lens]
    w = np.zeros((K, in_dim, cout), np.float32)
    w[:, :feat] = rng.standard_normal((K, feat, cout)) / np.sqrt(K * feat)
    layout = engine.BatchLayout(lens, 2, 8)
    host = np.zeros((layout.rows, in_dim), np.float32)
    layout.pack(mats, host)
    R = layout.rows - 5                                            # not a multiple of 16
    t = lambda a: torch.from_numpy(a).to(dev)
    x, rv = t(host), t(layout.row_valid())
    b = t((0.1 * rng.standard_normal(cout)).astype(np.float32))
    scale, shift = hiplib.fold_bn(*(t(a) for a in _rand_bn(rng, cout)), 1e-3)
    first = hiplib.pack_first_bf16x3(t(w))
    f = hiplib.FMT_SPLIT8 if fmt == "split8" else hiplib.FMT_SPLIT
    status = torch.zeros(1, dtype=torch.int32, device=dev)

    def run(n):
        y = hiplib.SplitBuf(layout.rows, cout, dev, f)
        y.base.fill_(0x7f)
        hiplib.set_tuning(hiplib.TUNE_FIRST_TILES, n)
        try:
            hiplib.tdnn_first(x, R, first, b, scale, shift, 1, None, dil, rv, y, status if f == hiplib.FMT_SPLIT8 else None)
            torch.cuda.synchronize()
        finally:
            hiplib.set_tuning(hiplib.TUNE_FIRST_TILES, 0)
        return y.base.cpu().numpy().copy(), hiplib.split_decode(y, layout.rows).cpu().numpy()
    raw, got = run(tiles)
    raw1, _ = run(4)
    assert np.array_equal(raw, raw1)                               # the run length changes nothing, poison behind row R included
    y2 = hiplib.SplitBuf(layout.rows, cout, dev)
    hiplib.tdnn_layer(x, hiplib.pack_weights_bf16x3(t(w)), b, scale, shift, 1, None, K, dil, rv, y2, None, rows=R)
    two = hiplib.split_decode(y2, layout.rows).cpu().numpy()
    assert np.isfinite(got[:R]).all()
    assert oracle.rel_l2(got[:R], two[:R]) < (2e-5 if fmt == "split8" else 5e-6)
    assert (got[:R][~layout.row_valid().astype(bool)[:R]] == 0).all()
    assert int(status.item()) == 0


def test_tdnn_first_layer_beyond_one_buffer_descriptor(env):
    """The first-layer kernel addresses y through a buffer descriptor (rows past the end are dropped by its range check), which
    covers 2^19 rows per launch: 2^19 + 1000 rows go out as two launches; against the general kernel, and the last rows exactly."""
    torch, hiplib, oracle, dev = env["torch"], env["hiplib"], env["oracle"], env["dev"]
    K, in_dim, cout = 5, 24, 64
    R = (1 << 19) + 1000
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((R, in_dim), generator=g).mul_(3.0)
    x[:, 23] = 0
    x = x.to(dev)
    w = (torch.randn((K, in_dim, cout), generator=g) / (K * 23) ** 0.5).to(dev)
    b = (0.1 * torch.randn(cout, generator=g)).to(dev)
    rv = torch.ones(R, dtype=torch.uint8, device=dev)
    rv[(1 << 19) - 3:(1 << 19) + 2] = 0                            # gap rows across the seam
    first = hiplib.pack_first_bf16x3(w)
    y = hiplib.SplitBuf(R, cout, dev)
    y.base.fill_(0x7f)
    hiplib.tdnn_first(x, R, first, b, None, None, 1, None, 1, rv, y)
    got = hiplib.split_decode(y, R)
    y2 = hiplib.SplitBuf(R, cout, dev)
    hiplib.tdnn_layer(x, hiplib.pack_weights_bf16x3(w), b, None, None, 1, None, K, 1, rv, y2, None, rows=R)
    two = hiplib.split_decode(y2, R)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got).all())
    num = float(((got.double() - two.double()) ** 2).sum().sqrt()), float((two.double() ** 2).sum().sqrt())
    assert num[0] / num[1] < 5e-6
    seam = slice((1 << 19) - 40, (1 << 19) + 40)
    assert oracle.rel_l2(got[seam].cpu().numpy(), two[seam].cpu().numpy()) < 5e-6
    assert oracle.rel_l2(got[-50:].cpu().numpy(), two[-50:].cpu().numpy()) < 5e-6
    assert bool((got[(1 << 19) - 3:(1 << 19) + 2] == 0).all())


def test_tdnn_first_layer_clamps_and_flags_out_of_range_values(env):
    """split8 output of the first layer: a value beyond +-57344 is clamped and raises the status word (the kernel starts on a form
    without the clamp and repeats the tile pair with it once its running maximum leaves the range); everything else is untouched."""
    torch, hiplib, dev = env["torch"], env["hiplib"], env["dev"]
    K, in_dim, cout = 5, 24, 64
    R = 700
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randn((R, in_dim), generator=g)
    x[:, 23] = 0
    x = x.to(dev)
    w = (torch.randn((K, in_dim, cout), generator=g) / (K * 23) ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev)
    scale = torch.ones(cout, device=dev)
    shift = torch.zeros(cout, device=dev)
    rv = torch.ones(R, dtype=torch.uint8, device=dev)
    first = hiplib.pack_first_bf16x3(w)

    def run(sc):
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        y = hiplib.SplitBuf(R, cout, dev, hiplib.FMT_SPLIT8)
        hiplib.tdnn_first(x, R, first, b, sc, shift, 1, None, 1, rv, y, status)
        return hiplib.split_decode(y, R).cpu().numpy(), int(status.item())
    base, st0 = run(scale)
    assert st0 == 0 and np.abs(base).max() < 100
    big = scale.clone()
    big[37] = 1e6                                                  # relu output of channel 37 x 1e6: far beyond the range
    got, st1 = run(big)
    assert st1 == 1
    hot = base[:, 37] * 1e6 > 57344
    assert hot.any() and (got[hot, 37] == 57344.0).all()
    keep = np.ones(cout, bool); keep[37] = False
    assert np.array_equal(got[:, keep], base[:, keep])
    small = base[:, 37] * 1e6 < 50000
    assert np.allclose(got[small, 37], base[small, 37] * 1e6, rtol=2e-3)


@pytest.mark.parametrize("cin,cout,act,lens", [
    (512, 1536, "relu", [25, 1, 7, 8, 9, 130, 257, 1000]),      # layers 3 + 4 of the default topology
    (64, 64, "prelu", [300, 25, 64, 3]),                         # two k-steps, one column group
    (96, 192, "lrelu", [1200, 33]),                              # long chunk: many blocks per chunk
    (512, 1536, "none", [40, 129]),
])
def test_tdnn_pair_pool_matches_oracle(env, cin, cout, act, lens):
    """xv_tdnn_pair_pool_bf16x3 (two K = 1 layers chained in registers + pooling block statistics) +
    xv_stats_pool_blocks_f32 == statistics pooling of layer(layer(x)) in the fp64 oracle; and the same block statistics
    as the two-launch path (xv_tdnn_layer_bf16x3 -> xv_tdnn_layer_pool_bf16x3) up to fp32 summation order."""
    torch, hiplib, engine, oracle, dev = env["torch"], env["hiplib"], env["engine"], env["oracle"], env["dev"]
    cmid = 512
    assert hiplib.pair_supported(cin, cmid, cout) and not hiplib.pair_supported(cin, 256, cout) and not hiplib.pair_supported(40, cmid, cout)
    rng = np.random.default_rng(cin + cout + len(lens))
    mats = [(rng.standard_normal((t, cin)) * 2).astype(np.float32) for t in lens]
    w1 = (rng.standard_normal((1, cin, cmid)) / np.sqrt(cin)).astype(np.float32)
    w2 = (rng.standard_normal((1, cmid, cout)) / np.sqrt(cmid)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(cmid)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    bn1, bn2 = _rand_bn(rng, cmid), _rand_bn(rng, cout)
    a1 = a2 = None
    if act == "lrelu":
        a1 = a2 = np.array([0.2], np.float32)
    elif act == "prelu":
        a1 = (0.1 + 0.05 * rng.standard_normal(cmid)).astype(np.float32)
        a2 = (0.1 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
    layout = engine.BatchLayout(lens, 1, hiplib.POOL_BLOCK_ROWS)
    host = np.zeros((layout.rows, cin), np.float32)
    layout.pack(mats, host)
    x = torch.from_numpy(host).to(dev)
    rv = torch.from_numpy(layout.row_valid()).to(dev)
    xin = hiplib.SplitBuf(layout.rows, cin, dev)
    hiplib.split_encode(x, xin)
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    s1, o1 = hiplib.fold_bn(*(t(a) for a in bn1), 1e-3)
    s2, o2 = hiplib.fold_bn(*(t(a) for a in bn2), 1e-3)
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]
    pair = hiplib.pack_pair_bf16x3(t(w1[0]), t(w2[0]))
    blk = torch.full((hiplib.block_stats_floats(layout.rows, cout),), float("nan"), dtype=torch.float32, device=dev)
    hiplib.tdnn_pair_pool(xin, layout.rows, pair, (t(b1), s1, o1, t(a1)), (t(b2), s2, o2, t(a2)), code, rv, blk)
    out = torch.full((len(lens), 2 * cout), float("nan"), dtype=torch.float32, device=dev)
    rs, rl = torch.from_numpy(layout.row_start).to(dev), torch.from_numpy(layout.row_len).to(dev)
    hiplib.stats_pool_blocks(blk, cout, rs, rl, len(lens), 1e-5, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    for i, m in enumerate(mats):
        h = oracle.tdnn_layer(m, w1, b1, bn1, act, a1, 1, np.float64)
        ref = oracle.stats_pool(oracle.tdnn_layer(h, w2, b2, bn2, act, a2, 1, np.float64), 1e-5, np.float64)
        assert oracle.rel_l2(got[i, :cout], ref[:cout]) < 2 * TOL_GEMM3, (i, lens[i])
        assert oracle.rel_l2(got[i, cout:], ref[cout:]) < 2 * TOL_GEMM3, (i, lens[i])
    # the two-launch path on the same input: same quantity, different fp32 summation order
    hmid = hiplib.SplitBuf(layout.rows, cmid, dev)
    hiplib.tdnn_layer(xin, hiplib.pack_weights_bf16x3(t(w1)), t(b1), s1, o1, code, t(a1), 1, 1, rv, hmid, None, rows=layout.rows)
    blk2 = torch.full_like(blk, float("nan"))
    hiplib.tdnn_layer_pool(hmid, layout.rows, hiplib.pack_weights_bf16x3(t(w2)), t(b2), s2, o2, code, t(a2), 1, rv, blk2)
    out2 = torch.empty_like(out)
    hiplib.stats_pool_blocks(blk2, cout, rs, rl, len(lens), 1e-5, out2)
    torch.cuda.synchronize()
    two = out2.cpu().numpy()
    for i in range(len(lens)):
        assert oracle.rel_l2(got[i], two[i]) < 1e-5, (i, lens[i])
    # blocks beyond the last row are never written; every block that holds frames is
    b1_, b2_ = blk.cpu().numpy().reshape(-1, 2, cout), blk2.cpu().numpy().reshape(-1, 2, cout)
    has_frames = layout.row_valid().astype(bool)
    has_frames = np.pad(has_frames, (0, (-len(has_frames)) % 8)).reshape(-1, 8).any(axis=1)
    assert np.isfinite(b1_[has_frames]).all() and np.allclose(b1_[has_frames], b2_[has_frames], rtol=2e-3, atol=2e-4)
    # batch composition does not change a chunk's bits: chunk 0 alone
    lay1 = engine.BatchLayout(lens[:1], 1, hiplib.POOL_BLOCK_ROWS)
    host1 = np.zeros((lay1.rows, cin), np.float32)
    lay1.pack(mats[:1], host1)
    x1 = hiplib.SplitBuf(lay1.rows, cin, dev)
    hiplib.split_encode(torch.from_numpy(host1).to(dev), x1)
    blk1 = torch.full((hiplib.block_stats_floats(lay1.rows, cout),), float("nan"), dtype=torch.float32, device=dev)
    hiplib.tdnn_pair_pool(x1, lay1.rows, pair, (t(b1), s1, o1, t(a1)), (t(b2), s2, o2, t(a2)), code,
                          torch.from_numpy(lay1.row_valid()).to(dev), blk1)
    out1 = torch.empty((1, 2 * cout), dtype=torch.float32, device=dev)
    hiplib.stats_pool_blocks(blk1, cout, torch.from_numpy(lay1.row_start).to(dev), torch.from_numpy(lay1.row_len).to(dev), 1, 1e-5, out1)
    torch.cuda.synchronize()
    assert np.array_equal(out1.cpu().numpy()[0], got[0])


def test_stats_pool_constant_channel_is_exact(env):
    """A dead-ReLU channel is constant over time: mean exact, std == sqrt(eps) exactly."""
    torch, hiplib, dev = env["torch"], env["hiplib"], env["dev"]
    T, C = 300, 64
    host = np.tile(np.linspace(-2, 2, C).astype(np.float32), (T, 1))
    h = torch.from_numpy(host).to(dev)
    rs = torch.zeros(1, dtype=torch.int32, device=dev)
    rl = torch.full((1,), T, dtype=torch.int32, device=dev)
    out = torch.empty((1, 2 * C), dtype=torch.float32, device=dev)
    hiplib.stats_pool(h, rs, rl, 1, T, 512, 1e-5, out, None)
    got = out.cpu().numpy()[0]
    assert np.array_equal(got[:C], host[0])
    assert np.array_equal(got[C:], np.full(C, np.sqrt(np.float32(1e-5)), np.float32))


def test_fc_matches_oracle(env):
    torch, hiplib, oracle, dev = env["torch"], env["hiplib"], env["oracle"], env["dev"]
    rng = np.random.default_rng(3)
    B, In, Out = 77, 3072, 512
    x = rng.standard_normal((B, In)).astype(np.float32)
    w = (rng.standard_normal((In, Out)) / np.sqrt(In)).astype(np.float32)
    b = (0.1 * rng.standard_normal(Out)).astype(np.float32)
    bn = _rand_bn(rng, Out)
    xd = torch.from_numpy(x).to(dev)
    wp = hiplib.pack_weights(torch.from_numpy(w).to(dev))
    assert np.array_equal(wp.cpu().numpy(), w.T)
    scale, shift = hiplib.fold_bn(*(torch.from_numpy(a).to(dev) for a in bn), 1e-3)
    y = torch.empty((B, Out), dtype=torch.float32, device=dev)
    ypre = torch.empty_like(y)
    hiplib.fc(xd, wp, torch.from_numpy(b).to(dev), scale, shift, 1, None, y, ypre)
    torch.cuda.synchronize()
    ref_pre = oracle.fc(x, w, b, np.float64)
    ref = oracle.act_bn(ref_pre, bn, "relu", None, np.float64)
    assert oracle.rel_l2(ypre.cpu().numpy(), ref_pre) < TOL_GEMM
    assert oracle.rel_l2(y.cpu().numpy(), ref) < TOL_GEMM
    hiplib.fc(xd, hiplib.pack_weights_bf16x3(torch.from_numpy(w[None]).to(dev)), torch.from_numpy(b).to(dev), scale, shift, 1,
              None, y, ypre)
    torch.cuda.synchronize()
    assert oracle.rel_l2(ypre.cpu().numpy(), ref_pre) < TOL_GEMM3
    assert oracle.rel_l2(y.cpu().numpy(), ref) < TOL_GEMM3


def test_chunk_average_bit_exact(env):
    torch, hiplib, oracle, dev = env["torch"], env["hiplib"], env["oracle"], env["dev"]
    rng = np.random.default_rng(9)
    D = 512
    segs = [1, 3, 1, 4, 2]
    lens = [rng.integers(25, 10001, size=n).astype(np.int32) for n in segs]
    embs = [rng.standard_normal((n, D)).astype(np.float32) * 5 for n in segs]
    e = torch.from_numpy(np.concatenate(embs)).to(dev)
    seg = torch.tensor(np.concatenate([[0], np.cumsum(segs)]), dtype=torch.int32, device=dev)
    cl = torch.from_numpy(np.concatenate(lens)).to(dev)
    out = torch.empty((len(segs), D), dtype=torch.float32, device=dev)
    hiplib.chunk_average(e, seg, cl, len(segs), out)
    got = out.cpu().numpy()
    for i in range(len(segs)):
        ref = oracle.chunk_average(embs[i], lens[i], np.float32)
        assert np.array_equal(got[i], ref), i
        # and the NumPy expression of the reference itself (models.py:398,418-421)
        acc, tot = 0, 0.0
        for ln, ev in zip(lens[i], embs[i]):
            tot += int(ln)
            acc = acc + int(ln) * ev
        acc = acc / tot
        assert np.array_equal(got[i], acc.astype(np.float32))


def test_bad_arguments_fail_loudly(env):
    torch, hiplib, dev = env["torch"], env["hiplib"], env["dev"]
    x = torch.zeros((10, 8), dtype=torch.float32, device=dev)
    wp = torch.zeros((8, 9 * 8), dtype=torch.float32, device=dev)
    y = torch.zeros((10, 8), dtype=torch.float32, device=dev)
    with pytest.raises(hiplib.XvectorHipError):
        hiplib.tdnn_layer(x, wp, None, None, None, 1, None, 9, 2, None, y)      # (K-1)*dil = 16 > 8


# ---- self-attentive pooling kernels (models.py:1036-1050) ----------------------------------------------------------------
@pytest.mark.parametrize("A,lens,split,strided", [
    (1536, [25, 200, 400, 1, 7, 33, 512], 512, True),      # direct path; h2 = right half of a [R, 2A] buffer
    (1536, [25, 2000, 513, 10000, 100], 512, False),       # split path + merge kernel
    (48, [5, 64, 300], 512, True),                         # A < 64: partially filled wave
    (256, [1000, 999], 128, False),
])
def test_attention_scores_softmax_pool_match_oracle(env, A, lens, split, strided):
    torch, hiplib, engine, oracle, dev = env["torch"], env["hiplib"], env["engine"], env["oracle"], env["dev"]
    rng = np.random.default_rng(A + sum(lens))
    layout = engine.BatchLayout(lens, 3)
    R = layout.rows
    u_host = (1.5 * rng.standard_normal((R, A))).astype(np.float32)
    v_host = (rng.standard_normal(A) * (2.0 / np.sqrt(A))).astype(np.float32)
    full = (rng.standard_normal((R, 2 * A)) * 1.7 + 3.0 * rng.standard_normal(2 * A)).astype(np.float32)
    full[: layout.lead] = 1e6                                   # garbage in the gaps must not matter
    u, v = torch.from_numpy(u_host).to(dev), torch.from_numpy(v_host).to(dev)
    hbuf = torch.from_numpy(full).to(dev)
    h2 = hbuf[:, A:] if strided else hbuf[:, A:].contiguous()
    rs, rl = torch.from_numpy(layout.row_start).to(dev), torch.from_numpy(layout.row_len).to(dev)
    scores = torch.empty(R, dtype=torch.float32, device=dev)
    nl = torch.empty((R, A), dtype=torch.float32, device=dev)
    att = torch.full((R,), float("nan"), dtype=torch.float32, device=dev)
    out = torch.full((len(lens), 2 * A), float("nan"), dtype=torch.float32, device=dev)
    need = hiplib.attention_pool_workspace_bytes(A, len(lens), max(lens), split)
    assert (need > 0) == (max(lens) > split)
    ws = torch.empty(max(need // 8, 1), dtype=torch.float64, device=dev)
    hiplib.attention_scores(u, v, scores, nl)
    hiplib.attention_softmax(scores, rs, rl, len(lens), att)
    hiplib.attention_pool(h2, att, rs, rl, len(lens), max(lens), split, 1e-5, out, ws)
    torch.cuda.synchronize()
    s_ref = np.tanh(u_host.astype(np.float64)) @ v_host.astype(np.float64)
    assert np.abs(scores.cpu().numpy() - s_ref).max() < 2e-6 * max(1.0, np.abs(s_ref).max())
    assert np.abs(nl.cpu().numpy() - np.tanh(u_host.astype(np.float64))).max() < 5e-7
    got, a_got = out.cpu().numpy(), att.cpu().numpy()
    for i, (s, n) in enumerate(zip(layout.row_start, layout.row_len)):
        sl = slice(int(s), int(s) + int(n))
        e = np.exp(s_ref[sl] - s_ref[sl].max())
        a = e / e.sum()
        assert np.abs(a_got[sl] - a).max() < 1e-5 * a.max()
        x = full[sl, A:].astype(np.float64)
        m = a @ x
        sd = np.sqrt(a @ (x * x) - m * m + 1e-5)
        assert oracle.rel_l2(got[i, :A], m) < 1e-5 and oracle.rel_l2(got[i, A:], sd) < 1e-5, (i, n)
    assert np.isnan(a_got[: layout.lead]).all()                 # gap rows are never written


def test_attention_pool_uniform_weights_equal_plain_statistics(env):
    """With constant scores the attention is 1/T and the weighted moments are tf.nn.moments: the two pooling kernels agree."""
    torch, hiplib, engine, dev = env["torch"], env["hiplib"], env["engine"], env["dev"]
    rng = np.random.default_rng(3)
    lens, A = [300, 25, 1111], 192
    layout = engine.BatchLayout(lens, 3)
    h = torch.from_numpy((rng.standard_normal((layout.rows, A)) + 2.0).astype(np.float32)).to(dev)
    rs, rl = torch.from_numpy(layout.row_start).to(dev), torch.from_numpy(layout.row_len).to(dev)
    scores = torch.full((layout.rows,), 0.7, dtype=torch.float32, device=dev)
    att = torch.zeros(layout.rows, dtype=torch.float32, device=dev)
    a_out = torch.empty((3, 2 * A), dtype=torch.float32, device=dev)
    p_out = torch.empty((3, 2 * A), dtype=torch.float32, device=dev)
    ws = torch.empty(hiplib.attention_pool_workspace_bytes(A, 3, max(lens), 512) // 8 + 1, dtype=torch.float64, device=dev)
    ws2 = torch.empty(hiplib.stats_pool_workspace_bytes(A, 3, max(lens), 512) // 4 + 1, dtype=torch.float32, device=dev)
    hiplib.attention_softmax(scores, rs, rl, 3, att)
    hiplib.attention_pool(h, att, rs, rl, 3, max(lens), 512, 1e-5, a_out, ws)
    hiplib.stats_pool(h, rs, rl, 3, max(lens), 512, 1e-5, p_out, ws2)
    torch.cuda.synchronize()
    assert torch.allclose(a_out, p_out, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("nrows,in_dim,out_dim,act", [(64, 3072, 512, "relu"), (7, 1024, 200, "lrelu"), (128, 608, 64, "none"), (64, 256, 512, "prelu")])
def test_fc_splitk_matches_oracle_and_the_plain_fc(env, nrows, in_dim, out_dim, act):
    """xv_fc_splitk_f32 (the training minibatch's skinny segment-level FC: reduction dealt to groups of workgroups, groups added in
    order): against the float64 product, within fp32 rounding of xv_fc_f32, deterministic; a shape that is not skinny (the last
    one: 8 slabs) IS xv_fc_f32, bit for bit."""
    torch, hiplib, dev = env["torch"], env["hiplib"], env["dev"]
    rng = np.random.default_rng(nrows + in_dim)
    x = rng.standard_normal((nrows, in_dim)).astype(np.float32)
    w = (rng.standard_normal((in_dim, out_dim)) / np.sqrt(in_dim)).astype(np.float32)
    b = (0.1 * rng.standard_normal(out_dim)).astype(np.float32)
    alpha = np.array([0.2], np.float32) if act == "lrelu" else (0.1 + 0.05 * rng.standard_normal(out_dim)).astype(np.float32) if act == "prelu" else None
    code = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}[act]
    xd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
    wp = hiplib.pack_weights(torch.from_numpy(w).to(dev))
    al = None if alpha is None else torch.from_numpy(alpha).to(dev)
    outs = []
    for fn in (hiplib.fc_splitk, hiplib.fc_splitk, lambda *a: hiplib.fc(*a)):
        y = torch.full((nrows, out_dim), float("nan"), device=dev)
        z = torch.full((nrows, out_dim), float("nan"), device=dev)
        fn(xd, wp, bd, None, None, code, al, y, z)
        torch.cuda.synchronize()
        outs.append((y.cpu().numpy(), z.cpu().numpy()))
    zref = x.astype(np.float64) @ w.astype(np.float64) + b
    a = 0.0 if act in ("none", "relu") else alpha
    yref = zref if act == "none" else np.maximum(zref, 0) + a * np.minimum(zref, 0)
    rel = lambda g, r: np.linalg.norm(g - r) / np.linalg.norm(r)
    assert rel(outs[0][1], zref) < TOL_GEMM and rel(outs[0][0], yref) < TOL_GEMM
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])          # deterministic
    assert rel(outs[0][1], outs[2][1]) < 3e-6          # (the plain form's 3072-term fp32 chain is itself ~1e-6 from fp64)
    skinny = hiplib.fc_splitk_supported(nrows, in_dim, out_dim)
    assert skinny == (in_dim >= 512)
    if not skinny:
        assert np.array_equal(outs[0][1], outs[2][1])
