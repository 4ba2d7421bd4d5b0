"""CPU oracle for the x-vector extraction hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / reported baseline.  The product package never
imports it.

It restates, on the CPU, what the reference (BUTSpeechFIT/x-vector-kaldi-tf) computes on the
extraction path; file:line citations are relative to the reference root:

* forward graph            local/tf/models.py:50-94 (== 466-500 ModelWithoutDropout,
                           569-605 ModelWithoutDropoutTdnn; attention pooling 1036-1050)
* batch-norm eval          local/tf/tf_block.py:9-16,25-28
* which tensor is the xvec local/tf/models.py:159,414 (``embed_layer-0/scores:0``)
* chunk / average driver   local/tf/models.py:373-423

Two implementations live here: a thin ctypes binding of ``libxv_oracle.so`` (plain C,
``oracle/xv_oracle.c``; fp32 and fp64) and an independent pure-NumPy fp64 version
(``forward_numpy``) used to cross-check the C one.

PARITY STATUS (details in the header of ``oracle/xv_oracle.c``): ``forward`` / ``forward_numpy`` / ``attention_pool`` /
``embed_utterance`` are pinned against the reference's own graph code EXECUTED -- every ``build_model`` + ``load_model`` +
``make_embedding`` of local/tf/models.py run under tests/golden/numpy_tf1.py; tests/golden/forward_refgraph.npz holds what they
return and tests/test_oracle.py::test_oracle_matches_the_reference_graph requires < 1e-12 on all 8 classes, every layer, the pooled
vector and both embeddings.  Statements here that are now reference-executed: layer op order (bias -> activation -> BN), BN_EPSILON,
VAR2STD_EPSILON, SAME padding incl. dilation, the PReLU / leaky-ReLU forms, the attention split and softmax axis, which tensors
embedding[0] / [1] are, the variable names.  What stays by definition: the numerics of each TF op (TensorFlow itself is absent),
and ``sliding_cmn`` / ``select_voiced`` (Kaldi is absent: PARITY UNPINNED, said where they are defined).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BN_EPSILON = 1e-3          # tf_block.py:9  (batch_norm_wrapper default epsilon)
VAR2STD_EPSILON = 1e-5     # models.py:16

DEFAULT_TOPOLOGY = dict(          # models.py:27-29
    layer_sizes=[512, 512, 512, 512, 1536],
    kernel_sizes=[5, 5, 7, 1, 1],
    dilations=[1, 1, 1, 1, 1],
    embedding_sizes=[512, 512],
    activation="relu",
    lrelu_alpha=0.2,              # models.py:912
)
DILATED_TOPOLOGY = dict(DEFAULT_TOPOLOGY, kernel_sizes=[5, 3, 3, 1, 1],    # models.py:545-548
                        dilations=[1, 2, 3, 1, 1])

ATTENTION_TOPOLOGY = dict(DEFAULT_TOPOLOGY, layer_sizes=[512, 512, 512, 512, 3072], activation="lrelu",   # models.py:992-994
                          pooling="attention")

_ACT = {"none": 0, "relu": 1, "lrelu": 2, "prelu": 3}


def build(force=False):
    """Compile libxv_oracle.so with the committed Makefile (gcc only)."""
    so = os.path.join(_HERE, "libxv_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("xv_oracle.c", "xv_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libxv_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def _sfx(dt):
    return "_f32" if np.dtype(dt) == np.float32 else "_f64"


def tdnn_layer(x, w, b, bn, act="relu", alpha=None, dilation=1, dtype=np.float64):
    """One frame-level layer on ONE utterance.  bn = (gamma, beta, mean, var) or None."""
    dt = np.dtype(dtype)
    x = _c(x, dt); w = _c(w, dt); b = _c(b, dt)
    T, Cin = x.shape
    K, Cin2, Cout = w.shape
    assert Cin == Cin2
    y = np.empty((T, Cout), dt)
    g = be = m = v = None
    if bn is not None:
        g, be, m, v = (_c(a, dt) for a in bn)
    al = _c(np.atleast_1d(alpha), dt) if alpha is not None else None
    fn = getattr(lib(), "xv_oracle_tdnn_layer" + _sfx(dt))
    fn(_p(x), ctypes.c_int(T), ctypes.c_int(Cin), _p(w), _p(b), _p(g), _p(be), _p(m), _p(v),
       ctypes.c_double(BN_EPSILON), ctypes.c_int(0 if bn is None else 1),
       ctypes.c_int(_ACT[act]), _p(al), ctypes.c_int(K), ctypes.c_int(dilation),
       ctypes.c_int(Cout), _p(y))
    return y


def stats_pool(h, eps=VAR2STD_EPSILON, dtype=np.float64):
    dt = np.dtype(dtype)
    h = _c(h, dt)
    T, C = h.shape
    out = np.empty(2 * C, dt)
    getattr(lib(), "xv_oracle_stats_pool" + _sfx(dt))(_p(h), ctypes.c_int(T), ctypes.c_int(C),
                                                       ctypes.c_double(eps), _p(out))
    return out


def fc(x, w, b, dtype=np.float64):
    dt = np.dtype(dtype)
    x = _c(np.atleast_2d(x), dt); w = _c(w, dt); b = _c(b, dt)
    B, In = x.shape
    Out = w.shape[1]
    y = np.empty((B, Out), dt)
    getattr(lib(), "xv_oracle_fc" + _sfx(dt))(_p(x), ctypes.c_int(B), ctypes.c_int(In), _p(w), _p(b),
                                               ctypes.c_int(Out), _p(y))
    return y


def act_bn(x, bn, act="relu", alpha=None, dtype=np.float64):
    dt = np.dtype(dtype)
    x = _c(np.atleast_2d(x), dt)
    B, C = x.shape
    g, be, m, v = (_c(a, dt) for a in bn)
    al = _c(np.atleast_1d(alpha), dt) if alpha is not None else None
    y = np.empty_like(x)
    getattr(lib(), "xv_oracle_act_bn" + _sfx(dt))(_p(x), ctypes.c_int(B), ctypes.c_int(C), _p(g), _p(be),
                                                   _p(m), _p(v), ctypes.c_double(BN_EPSILON),
                                                   ctypes.c_int(_ACT[act]), _p(al), _p(y))
    return y


def _layer_alpha(weights, scope, topo):
    act = topo.get("activation", "relu")
    if act == "lrelu":
        return np.array([topo.get("lrelu_alpha", 0.2)])
    if act == "prelu":
        return weights["%s/prelu/prelu:0" % scope]       # tf_block.py:40-46
    return None


def _bn(weights, scope):
    return tuple(weights["%s/%s:0" % (scope, n)] for n in ("gamma", "beta", "mean", "variance"))


def attention_pool(h, weights, eps=VAR2STD_EPSILON, dtype=np.float64, return_attention=False):
    """Self-attentive statistics pooling of ModelL2LossWithoutDropoutLReluAttention (models.py:1036-1050), NumPy in ``dtype``:
    h[T, 2A] = [h1 | h2] (tf.split :1044); n = tanh(h1 W + b) (:1045); a = softmax_t(n v) (:1046);
    m = sum_t a h2 (:1048); q = sum_t a h2^2 - m^2 (:1049); pooled = [m | sqrt(q + 1e-5)] (:1050)."""
    h = np.asarray(h, dtype)
    A = h.shape[1] // 2
    h1, h2 = h[:, :A], h[:, A:]
    n = np.tanh(h1 @ np.asarray(weights["attention/w:0"], dtype) + np.asarray(weights["attention/b:0"], dtype))
    s = n @ np.asarray(weights["attention/v:0"], dtype)
    e = np.exp(s - s.max())
    a = (e / e.sum()).astype(dtype)
    m = a @ h2
    q = a @ (h2 * h2) - m * m
    pooled = np.concatenate([m, np.sqrt(q + dtype(eps))]).astype(dtype)
    return (pooled, a) if return_attention else pooled


def forward(x, weights, topo=None, dtype=np.float64, embedding_index=0, return_intermediates=False):
    """x[T,F] -> x-vector.  ``weights`` is keyed by the TF variable names (models.py:199-213)."""
    topo = topo or DEFAULT_TOPOLOGY
    act = topo.get("activation", "relu")
    h = np.asarray(x)
    inter = []
    for i, d in enumerate(topo["dilations"]):
        sc = "frame_level_info_layer-%d" % i
        h = tdnn_layer(h, weights[sc + "/w:0"], weights[sc + "/b:0"], _bn(weights, sc), act,
                       _layer_alpha(weights, sc, topo), d, dtype)
        inter.append(h)
    if topo.get("pooling", "stats") == "attention":
        pooled = attention_pool(h, weights, VAR2STD_EPSILON, dtype)
    else:
        pooled = stats_pool(h, VAR2STD_EPSILON, dtype)
    inter.append(pooled)
    e0 = fc(pooled[None, :], weights["embed_layer-0/w:0"], weights["embed_layer-0/b:0"], dtype)[0]
    inter.append(e0)
    out = e0
    if embedding_index == 1:
        a0 = act_bn(e0[None, :], _bn(weights, "embed_layer-0"), act,
                    _layer_alpha(weights, "embed_layer-0", topo), dtype)
        out = fc(a0, weights["embed_layer-1/w:0"], weights["embed_layer-1/b:0"], dtype)[0]
        inter.append(out)
    return (out, inter) if return_intermediates else out


def forward_numpy(x, weights, topo=None, embedding_index=0):
    """Independent pure-NumPy fp64 restatement (no C) -- cross-checks the C oracle."""
    topo = topo or DEFAULT_TOPOLOGY
    act = topo.get("activation", "relu")
    h = np.asarray(x, np.float64)

    def activation(z, scope):
        if act == "relu":
            return np.maximum(z, 0.0)
        if act == "lrelu":
            return np.maximum(topo.get("lrelu_alpha", 0.2) * z, z)
        if act == "prelu":
            a = np.asarray(weights["%s/prelu/prelu:0" % scope], np.float64)
            return np.maximum(0.0, z) + a * np.minimum(0.0, z)
        return z

    def bn(r, scope):
        g, be, m, v = (np.asarray(a, np.float64) for a in _bn(weights, scope))
        s = g / np.sqrt(v + BN_EPSILON)
        return r * s + (be - m * s)

    for i, (K, d) in enumerate(zip(topo["kernel_sizes"], topo["dilations"])):
        sc = "frame_level_info_layer-%d" % i
        w = np.asarray(weights[sc + "/w:0"], np.float64)
        T = h.shape[0]
        left = (K - 1) * d // 2
        hp = np.zeros((T + (K - 1) * d, h.shape[1]))
        hp[left:left + T] = h
        z = np.zeros((T, w.shape[2])) + np.asarray(weights[sc + "/b:0"], np.float64)
        for k in range(K):
            z += hp[k * d:k * d + T] @ w[k]
        h = bn(activation(z, sc), sc)
    if topo.get("pooling", "stats") == "attention":          # models.py:1036-1050, written out once more, independently
        A = h.shape[1] // 2
        u = h[:, :A] @ np.asarray(weights["attention/w:0"], np.float64) + np.asarray(weights["attention/b:0"], np.float64)
        s = np.tanh(u) @ np.asarray(weights["attention/v:0"], np.float64)
        a = np.exp(s - np.logaddexp.reduce(s))
        mu = np.einsum("tc,t->c", h[:, A:], a)
        var = np.einsum("tc,t->c", h[:, A:] ** 2, a) - mu ** 2
    else:
        mu = h.mean(axis=0)
        var = ((h - mu) ** 2).mean(axis=0)
    pooled = np.concatenate([mu, np.sqrt(var + VAR2STD_EPSILON)])
    e0 = pooled @ np.asarray(weights["embed_layer-0/w:0"], np.float64) + weights["embed_layer-0/b:0"]
    if embedding_index == 0:
        return e0
    a0 = bn(activation(e0, "embed_layer-0"), "embed_layer-0")
    return a0 @ np.asarray(weights["embed_layer-1/w:0"], np.float64) + weights["embed_layer-1/b:0"]


def chunk_plan(T, min_chunk_size, chunk_size):
    """[(start, len), ...] of the chunks make_embedding runs, or None if the key is rejected."""
    cap = max(1, T // max(1, (chunk_size if chunk_size > 0 else T) or 1) + 2)
    starts = (ctypes.c_int * cap)()
    lens = (ctypes.c_int * cap)()
    n = lib().xv_oracle_chunk_plan(ctypes.c_int(T), ctypes.c_int(min_chunk_size), ctypes.c_int(chunk_size),
                                   starts, lens, ctypes.c_int(cap))
    if n < 0:
        return None
    assert n <= cap
    return [(starts[i], lens[i]) for i in range(n)]


def chunk_average(embs, lens, dtype=np.float32):
    dt = np.dtype(dtype)
    e = _c(np.atleast_2d(embs), dt)
    ln = np.ascontiguousarray(lens, np.int32)
    out = np.empty(e.shape[1], dt)
    getattr(lib(), "xv_oracle_chunk_average" + _sfx(dt))(_p(e), _p(ln), ctypes.c_int(e.shape[0]),
                                                          ctypes.c_int(e.shape[1]), _p(out))
    return out


def embed_utterance(mat, weights, topo=None, min_chunk_size=25, chunk_size=10000, dtype=np.float64,
                    embedding_index=0):
    """One key of make_embedding (models.py:376-423).  Returns a float32 vector, or None when the
    reference writes nothing for this key.  Chunk embeddings are computed in ``dtype`` and rounded
    to float32 (what sess.run returns) before the float32 length-weighted average."""
    plan = chunk_plan(mat.shape[0], min_chunk_size, chunk_size)
    if plan is None:
        return None
    if not plan:
        # every chunk < min_chunk cannot happen when T >= min_chunk (first chunk is >= min); guard anyway
        return None
    embs = np.stack([forward(mat[s:s + n], weights, topo, dtype, embedding_index).astype(np.float32)
                     for s, n in plan])
    return chunk_average(embs, [n for _, n in plan], np.float32)


def rel_l2(a, ref):
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-300))


# ------------------------------------------------------------------------------------------------
# feature front-end: Kaldi's apply-cmvn-sliding / select-voiced-frames, which the reference calls as
# external binaries (local/tf/extract_xvectors.sh:68).  Kaldi is not vendored in the reference (version unpinned), so this
# restates the published algorithm of SlidingWindowCmn (kaldi/src/feat/feature-functions.cc): PARITY UNPINNED.
# ------------------------------------------------------------------------------------------------
def sliding_cmn(x, cmn_window=300, center=True, min_window=100):
    """Frame by frame as Kaldi does it: the window bounds, a running float64 window sum that adds / drops one frame at a
    time, out[t] = float32(float64(x[t]) - sum / frames).  norm_vars = false."""
    x64 = np.asarray(x, dtype=np.float64)
    T = x64.shape[0]
    out = np.empty(x64.shape, dtype=np.float32)
    last_start = last_end = -1
    cur = np.zeros(x64.shape[1], dtype=np.float64)
    for t in range(T):
        if center:
            ws = t - cmn_window // 2
            we = ws + cmn_window
        else:
            ws = t - cmn_window
            we = t + 1
        if ws < 0:
            we -= ws
            ws = 0
        if not center and we > t:
            we = max(t + 1, min_window)
        if we > T:
            ws -= we - T
            we = T
            if ws < 0:
                ws = 0
        if last_start == -1:
            cur = x64[ws:we].sum(axis=0)
        else:
            if ws > last_start:
                assert ws == last_start + 1
                cur = cur - x64[last_start]
            if we > last_end:
                assert we == last_end + 1
                cur = cur + x64[last_end]
        last_start, last_end = ws, we
        out[t] = (x64[t] - cur / float(we - ws)).astype(np.float32)
    return out


def select_voiced(feats, vad):
    """select-voiced-frames: rows whose VAD decision is non-zero; None when the lengths differ or nothing is voiced."""
    vad = np.asarray(vad).reshape(-1)
    if vad.shape[0] != feats.shape[0] or not np.any(vad != 0):
        return None
    return feats[vad != 0]
