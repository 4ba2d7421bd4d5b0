"""Seeded synthetic weights and utterances (data generation for tests and bench.py; there is no
network for checkpoints or corpora).  NOT a compute path: the BN calibration below is plain NumPy
used once to manufacture "trained-like" running statistics; the extractor never calls it.

* ``reference_init``      -- the reference's initialisers: ``truncated_normal(stddev=0.1)`` weights,
                             ``b = 0.1``, BN gamma=1 beta=0 mean=0 variance=1, Xavier output layer
                             (local/tf/models.py:56-58,82-84,98-100; local/tf/tf_block.py:10-14); He / Glorot
                             initialisers for ModelL2LossWithoutDropoutReluHeInit (models.py:1158-1210).
* ``trained_like``        -- the same shapes with fan-in-scaled weights, non-trivial gamma/beta and BN
                             running statistics calibrated on a random 2000-frame input, so that BN is
                             exercised and activations stay O(1) through the stack.
* ``make_utterances``     -- ``float32 [T,F]`` MFCC-like matrices, T uniform in [tmin, tmax].
* ``hostile``             -- weights a trained checkpoint COULD have and ``trained_like`` never produces: heavy-tailed
                             (Student-t, 3 degrees of freedom), per-channel scales spread over three decades, BN variances
                             from 1e-4 to 1e2, near-dead channels, wide gamma / beta -- the stress case of the reduced-
                             precision arithmetics (tests/test_gpu_hostile.py).
* ``mfcc_like``           -- utterances with the statistics of real cepstra after CMN: frame-to-frame AR(1) correlation,
                             a spectrum of per-coefficient scales, zero mean per utterance.
"""
import numpy as np

from . import topology as tp


def _truncated_normal(rng, shape, std):
    a = rng.standard_normal(size=shape)
    bad = np.abs(a) > 2.0
    while bad.any():
        a[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(a) > 2.0
    return (a * std).astype(np.float32)


def _shapes(topo, feat_dim):
    prev = feat_dim
    for i, (k, c) in enumerate(zip(topo["kernel_sizes"], topo["layer_sizes"])):
        yield "frame_level_info_layer-%d" % i, (k, prev, c)
        prev = c
    prev = tp.pooled_dim(topo)
    for j, c in enumerate(topo["embedding_sizes"]):
        yield "embed_layer-%d" % j, (prev, c)
        prev = c


def reference_init(topo, feat_dim, num_classes, seed=0):
    rng = np.random.default_rng(seed)
    w = {}
    last = feat_dim
    he = topo.get("init", "default") == "he"
    for scope, shape in _shapes(topo, feat_dim):
        c = shape[-1]
        if he:          # he_normal weights, he_uniform biases (models.py:1158-1163 frame level, 1181-1185 embedding layers)
            fan_in = int(np.prod(shape[:-1]))
            w[scope + "/w:0"] = _truncated_normal(rng, shape, np.sqrt(2.0 / fan_in))
            lim = np.sqrt(6.0 / fan_in)
            w[scope + "/b:0"] = rng.uniform(-lim, lim, size=c).astype(np.float32)
        else:
            w[scope + "/w:0"] = _truncated_normal(rng, shape, 0.1)
            w[scope + "/b:0"] = np.full(c, 0.1, np.float32)
        w[scope + "/gamma:0"] = np.ones(c, np.float32)
        w[scope + "/beta:0"] = np.zeros(c, np.float32)
        w[scope + "/mean:0"] = np.zeros(c, np.float32)
        w[scope + "/variance:0"] = np.ones(c, np.float32)
        if topo.get("activation") == "prelu":
            w[scope + "/prelu/prelu:0"] = np.full(c, 0.1, np.float32)     # tf_block.py:45-46
        last = c
    if tp.is_attention(topo):                                             # models.py:1040-1043
        a = topo["layer_sizes"][-1] // 2
        w["attention/b:0"] = np.full(a, 0.1, np.float32)
        w["attention/v:0"] = np.full(a, 0.1, np.float32)
        w["attention/w:0"] = _truncated_normal(rng, (a, a), 0.1)
    lim = np.sqrt(6.0 / (last + num_classes))                             # xavier_initializer (uniform)
    if he:              # glorot_normal weights, glorot_uniform bias (models.py:1205-1210)
        w["output/w:0"] = _truncated_normal(rng, (last, num_classes), np.sqrt(2.0 / (last + num_classes)))
        w["output/b:0"] = rng.uniform(-lim, lim, size=num_classes).astype(np.float32)
        return w
    w["output/w:0"] = rng.uniform(-lim, lim, size=(last, num_classes)).astype(np.float32)
    w["output/b:0"] = np.full(num_classes, 0.1, np.float32)
    return w


def _act(z, topo, alpha):
    a = topo.get("activation", "relu")
    if a == "relu":
        return np.maximum(z, 0.0)
    if a == "lrelu":
        return np.maximum(topo.get("lrelu_alpha", 0.2) * z, z)
    if a == "prelu":
        return np.maximum(z, 0.0) + alpha * np.minimum(z, 0.0)
    return z


def trained_like(topo, feat_dim, num_classes=64, seed=0, calib_frames=2000, input_scale=3.0):
    rng = np.random.default_rng(seed)
    w = {}
    h = (rng.standard_normal((calib_frames, feat_dim)) * input_scale)
    prev = feat_dim
    for i, (k, d, c) in enumerate(zip(topo["kernel_sizes"], topo["dilations"], topo["layer_sizes"])):
        sc = "frame_level_info_layer-%d" % i
        std = np.sqrt(2.0 / (k * prev))
        wt = _truncated_normal(rng, (k, prev, c), std)
        b = (0.1 + 0.05 * rng.standard_normal(c)).astype(np.float32)
        gamma = (1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32)
        beta = (0.1 * rng.standard_normal(c)).astype(np.float32)
        alpha = (0.1 + 0.02 * rng.standard_normal(c)).astype(np.float32)
        # calibration pass (float64 NumPy): z on the random input, zero padded as one utterance
        T = h.shape[0]
        left = (k - 1) * d // 2
        hp = np.zeros((T + (k - 1) * d, prev))
        hp[left:left + T] = h
        z = np.zeros((T, c)) + b.astype(np.float64)
        for kk in range(k):
            z += hp[kk * d:kk * d + T] @ wt[kk].astype(np.float64)
        r = _act(z, topo, alpha.astype(np.float64))
        mean = r.mean(0)
        var = r.var(0)
        # perturb the running stats a little: a trained net's moving averages never match one batch
        mean_f = (mean * (1.0 + 0.05 * rng.standard_normal(c))).astype(np.float32)
        var_f = (var * np.exp(0.1 * rng.standard_normal(c)) + 1e-4).astype(np.float32)
        s = gamma.astype(np.float64) / np.sqrt(var_f.astype(np.float64) + tp.BN_EPSILON)
        h = r * s + (beta.astype(np.float64) - mean_f.astype(np.float64) * s)
        w[sc + "/w:0"], w[sc + "/b:0"] = wt, b
        w[sc + "/gamma:0"], w[sc + "/beta:0"] = gamma, beta
        w[sc + "/mean:0"], w[sc + "/variance:0"] = mean_f, var_f
        if topo.get("activation") == "prelu":
            w[sc + "/prelu/prelu:0"] = alpha
        prev = c
    if tp.is_attention(topo):
        # scores with a standard deviation of ~1.5 over the frames: attention neither uniform nor one-hot
        a = prev // 2
        w["attention/w:0"] = _truncated_normal(rng, (a, a), np.sqrt(1.0 / a))
        w["attention/b:0"] = (0.1 * rng.standard_normal(a)).astype(np.float32)
        w["attention/v:0"] = (rng.standard_normal(a) * (1.5 / (0.6 * np.sqrt(a)))).astype(np.float32)
    prev = tp.pooled_dim(topo)
    for j, c in enumerate(topo["embedding_sizes"]):
        sc = "embed_layer-%d" % j
        w[sc + "/w:0"] = _truncated_normal(rng, (prev, c), np.sqrt(1.0 / prev))
        w[sc + "/b:0"] = (0.1 + 0.05 * rng.standard_normal(c)).astype(np.float32)
        w[sc + "/gamma:0"] = (1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32)
        w[sc + "/beta:0"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        w[sc + "/mean:0"] = (0.3 + 0.1 * rng.standard_normal(c)).astype(np.float32)
        w[sc + "/variance:0"] = np.exp(0.2 * rng.standard_normal(c)).astype(np.float32)
        if topo.get("activation") == "prelu":
            w[sc + "/prelu/prelu:0"] = (0.1 + 0.02 * rng.standard_normal(c)).astype(np.float32)
        prev = c
    lim = np.sqrt(6.0 / (prev + num_classes))
    w["output/w:0"] = rng.uniform(-lim, lim, size=(prev, num_classes)).astype(np.float32)
    w["output/b:0"] = np.full(num_classes, 0.1, np.float32)
    return w


def mfcc_like(lengths, feat_dim=23, seed=0, rho=0.92, c0_scale=12.0):
    """float32 [T, F] matrices shaped like MFCCs after sliding-window CMN: every coefficient an AR(1) process over the frames
    (``rho``: neighbouring frames of speech are strongly correlated -- the K taps of a TDNN layer see nearly the same vector),
    coefficient j scaled ``c0_scale / (1 + j)^0.7`` plus 5 % of its own scale shared with coefficient j-1 (cepstra of one frame
    are not independent), the utterance mean removed.  Occasional bursts (x4 for ~10 frames) stand in for non-speech events."""
    rng = np.random.default_rng(seed)
    scale = c0_scale / (1.0 + np.arange(feat_dim)) ** 0.7
    out = []
    for T in lengths:
        T = int(T)
        e = rng.standard_normal((T, feat_dim)) * np.sqrt(1.0 - rho * rho)
        x = np.empty((T, feat_dim))
        acc = rng.standard_normal(feat_dim)
        for t in range(T):
            acc = rho * acc + e[t]
            x[t] = acc
        x[:, 1:] += 0.05 * x[:, :-1]
        if T >= 40:
            for _ in range(max(1, T // 400)):
                t0 = int(rng.integers(0, T - 10))
                x[t0:t0 + 10] *= 4.0
        x *= scale
        x -= x.mean(axis=0, keepdims=True)
        out.append(x.astype(np.float32))
    return out


def hostile(topo, feat_dim, num_classes=64, seed=0, calib=None, decades=3.0, dead_fraction=0.06):
    """A weight set that computes a sane network (BN statistics are CALIBRATED on ``calib`` -- default: 3000 frames of
    ``mfcc_like`` input -- so activations stay finite and O(1) on such input) but has none of the comfortable properties of
    ``trained_like``:

    * weights ~ Student-t(3) (heavy tails: single weights 10-30x the typical one) at He scale;
    * every output channel of every layer multiplied by 10^U(-decades+1, 1): pre-activation variances -- and therefore the
      calibrated BN variances -- spread from ~1e-4 to ~1e2, BN scales ``gamma / sqrt(var + 1e-3)`` from ~0.1 to ~31;
    * ``dead_fraction`` of the channels near-dead: a bias 3-4 standard deviations below zero, the ReLU fires on a few frames;
    * gamma log-normal (sigma 0.7, a tenth of the channels another 30x down), beta ~ N(0, 0.5), and the INPUT side of the next
      layer compensating none of it.
    Returns the weight dictionary (same keys as ``trained_like``)."""
    rng = np.random.default_rng(seed)
    if calib is None:
        calib = np.concatenate(mfcc_like([1000, 1000, 1000], feat_dim, seed=seed + 99), axis=0)
    w = {}
    h = np.asarray(calib, dtype=np.float64)
    prev = feat_dim
    n = len(topo["kernel_sizes"])
    for i, (k, d, c) in enumerate(zip(topo["kernel_sizes"], topo["dilations"], topo["layer_sizes"])):
        sc = "frame_level_info_layer-%d" % i
        in_rms = np.sqrt((h * h).mean(axis=0)).mean() + 1e-12
        wt = rng.standard_t(3, size=(k, prev, c)) / np.sqrt(3.0) * np.sqrt(2.0 / (k * prev)) / in_rms
        ch = 10.0 ** rng.uniform(1.0 - decades, 1.0, size=c)
        wt = wt * ch[None, None, :]
        T = h.shape[0]
        left = (k - 1) * d // 2
        hp = np.zeros((T + (k - 1) * d, prev))
        hp[left:left + T] = h
        z0 = np.zeros((T, c))
        for kk in range(k):
            z0 += hp[kk * d:kk * d + T] @ wt[kk]
        zs = z0.std(axis=0) + 1e-30
        b = 0.3 * zs * rng.standard_normal(c)
        dead = rng.random(c) < dead_fraction
        b[dead] = -(3.0 + rng.random(int(dead.sum()))) * zs[dead] - z0.mean(axis=0)[dead]
        alpha = (0.1 + 0.05 * rng.standard_normal(c))
        r = _act(z0 + b, topo, alpha)
        mean, var = r.mean(0), r.var(0)
        gamma = np.exp(0.7 * rng.standard_normal(c))
        gamma[rng.random(c) < 0.1] /= 30.0
        if i == n - 1:
            gamma = np.exp(0.3 * rng.standard_normal(c))             # (what pooling sees keeps a common scale)
        beta = 0.5 * rng.standard_normal(c)
        mean_f = (mean * (1.0 + 0.05 * rng.standard_normal(c))).astype(np.float32)
        var_f = (var * np.exp(0.1 * rng.standard_normal(c))).astype(np.float32)
        sbn = gamma / np.sqrt(var_f.astype(np.float64) + tp.BN_EPSILON)
        h = r * sbn + (beta - mean_f.astype(np.float64) * sbn)
        w[sc + "/w:0"], w[sc + "/b:0"] = wt.astype(np.float32), b.astype(np.float32)
        w[sc + "/gamma:0"], w[sc + "/beta:0"] = gamma.astype(np.float32), beta.astype(np.float32)
        w[sc + "/mean:0"], w[sc + "/variance:0"] = mean_f, var_f
        if topo.get("activation") == "prelu":
            w[sc + "/prelu/prelu:0"] = alpha.astype(np.float32)
        prev = c
    base = trained_like(topo, feat_dim, num_classes, seed=seed + 1, calib_frames=64)
    for name, a in base.items():
        if not name.startswith("frame_level_info_layer-"):
            w[name] = a
    # the segment layer sees [mean | std] of hostile channels: Student-t weights there as well
    pooled = tp.pooled_dim(topo)
    w["embed_layer-0/w:0"] = (rng.standard_t(3, size=(pooled, topo["embedding_sizes"][0])) / np.sqrt(3.0) *
                              np.sqrt(1.0 / pooled)).astype(np.float32)
    return w


def utterance_lengths(n, tmin, tmax, seed=1234):
    rng = np.random.default_rng(seed)
    return rng.integers(tmin, tmax + 1, size=n).astype(np.int64)


def make_utterances(n, tmin, tmax, feat_dim=23, seed=1234, scale=3.0, key_fmt="utt%06d"):
    """[(key, float32[T,F])...] -- BASELINE configs: seed 1234, x ~ N(0,1)*3 (MFCC-like after CMVN)."""
    lens = utterance_lengths(n, tmin, tmax, seed)
    rng = np.random.default_rng(seed + 1)
    out = []
    for i, T in enumerate(lens):
        out.append((key_fmt % i, (rng.standard_normal((int(T), feat_dim)) * scale).astype(np.float32)))
    return out


SMALL_TOPOLOGY = dict(layer_sizes=[32, 32, 32, 32, 48], kernel_sizes=[5, 5, 7, 1, 1],
                      dilations=[1, 1, 1, 1, 1], embedding_sizes=[16, 16], activation="relu",
                      lrelu_alpha=0.2)


def speaker_minibatches(n_steps, feat_dim=23, n_spk=64, batch=64, tmin=200, tmax=400, seed=0, noise=3.0, spread=2.0):
    """Generator of ``(x float16 [batch, T, feat], labels int32 [batch])``: every speaker a mean vector ~ N(0, spread^2) under
    frame noise ~ N(0, noise^2), one length T ~ U{tmin..tmax} per minibatch (create_egs.py:508-513) -- the training input of
    BASELINE configs[4] and of ``trained_checkpoint`` below."""
    rng = np.random.default_rng(seed)
    spk = rng.standard_normal((n_spk, feat_dim)) * spread
    for _ in range(n_steps):
        T = int(rng.integers(tmin, tmax + 1))
        lab = rng.integers(0, n_spk, batch)
        yield (spk[lab][:, None, :] + rng.standard_normal((batch, T, feat_dim)) * noise).astype(np.float16), lab.astype(np.int32)


def trained_checkpoint(topo, feat_dim=23, n_spk=64, steps=300, learning_rate=1e-3, seed=0, device="cuda:0", precision="bf16x3"):
    """A checkpoint that was TRAINED, not sampled: reference-style initial weights (fan-in scaled), ``steps`` Adam steps of the
    product's own training step (xvector_amd/trainer.py, the twin of Model.train_one_iteration, local/tf/models.py:216-305) on
    ``speaker_minibatches`` -- the closest thing to a ``model_final`` (run_xvector.sh:88-107) an image without corpora can produce:
    the weights, biases and BatchNorm moving statistics are whatever the optimiser and the data made them, not draws chosen by the
    builder.  -> (weights {tf name: float32 ndarray}, dict(first_loss, last_loss, accuracy_last, seconds))."""
    import time
    from . import trainer
    w = reference_init(topo, feat_dim, n_spk, seed=seed)
    for k in list(w):                                         # fan-in scaled start so that activations stay O(1)
        if k.endswith("/w:0") and w[k].ndim == 3:
            w[k] = (w[k] * (np.sqrt(2.0 / (w[k].shape[0] * w[k].shape[1])) / 0.1)).astype(np.float32)
    tr = trainer.Trainer(w, topo, device, precision=precision)
    t0 = time.time()
    first = last = acc = None
    for x, lab in speaker_minibatches(steps, feat_dim, n_spk, seed=seed + 1):
        last, acc = tr.step(x, lab, learning_rate)
        if first is None:
            first = last
    out, _ = tr.export()
    return out, dict(first_loss=float(first), last_loss=float(last), accuracy_last=float(acc), seconds=time.time() - t0, steps=steps)
