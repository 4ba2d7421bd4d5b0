"""Training-step oracle (torch CPU, float64 autograd).  TEST INFRASTRUCTURE ONLY -- never imported by the product.

Restates the training graph of the reference (BUTSpeechFIT/x-vector-kaldi-tf) for the model classes without
dropout, following the TF op definitions at its call sites:

* forward, train phase          local/tf/models.py:466-500 (ModelWithoutDropout), 569-605 (Tdnn), 895-960 (LRelu), 1022-1083 (LReluAttention)
* batch-norm train branch       local/tf/tf_block.py:18-23: moments over every axis but the last (biased variance),
                                normalise with the BATCH statistics, moving <- moving*decay + batch*(1-decay), decay 0.95
                                (models.py:65,482)
* loss / accuracy               local/tf/models.py:106-117: mean softmax cross-entropy, argmax accuracy
* optimiser                     tf.train.AdamOptimizer(learning_rate) defaults beta1=0.9 beta2=0.999 epsilon=1e-8
                                (models.py:112): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMAs; var -= lr_t*m/(sqrt(v)+eps)
* eval phase                    same graph with the moving statistics (tf_block.py:25-26), models.py:307-354

PARITY STATUS: pinned against the reference's own training loop EXECUTED: ``train_one_iteration`` (models.py:216-305) and ``eval``
(:307-354) of all 8 classes run under tests/golden/numpy_tf1.py on graphs their own build_model made (batch-norm train branch, decay,
the L2 terms, class Model's dropout sites with the masks it drew, AdamOptimizer with its slots saved and restored through the
checkpoint); tests/golden/train_refgraph.npz holds three steps' losses, the step-0 gradients, weights / moving statistics / Adam slots
after the steps and the eval losses, and tests/test_oracle.py::test_training_oracle_matches_the_reference_graph requires this module
to reproduce them to 1e-9.  The numerics of each TF op and of reverse mode are numpy_tf1's (documented definitions, checked against
finite differences in tests/test_numpy_tf1.py); TensorFlow itself never runs.  The AM-softmax head is build-defined (not in the
reference): unpinned.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPSILON = 1e-3
VAR2STD_EPSILON = 1e-5
BN_DECAY = 0.95
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8


def trainable_names(topo):
    names = []
    scopes = ["frame_level_info_layer-%d" % i for i in range(len(topo["layer_sizes"]))] + \
             ["embed_layer-%d" % j for j in range(len(topo["embedding_sizes"]))]
    for sc in scopes:
        names += ["%s/%s:0" % (sc, n) for n in ("w", "b", "gamma", "beta")]
        if topo.get("activation") == "prelu":
            names.append("%s/prelu/prelu:0" % sc)
    if topo.get("pooling", "stats") == "attention":
        names += ["attention/w:0", "attention/b:0", "attention/v:0"]     # models.py:1040-1043
    return names + ["output/w:0", "output/b:0"]


# Every activation here has a kink at z = 0.  A float32 implementation whose pre-activation differs from the float64 one by its rounding
# error (~1e-6 of the layer's rms) can land on the other side of it, and ONE such element changes a layer's gradients by 1 / sqrt(elements)
# (3e-3 for 8 x 156 x 88) -- a property of the comparison, not an error of either side.  ``kink_margin`` lets a test tell: the smallest
# |z| / rms(z) over the activation sites of the last forward() call.
kink_margin = [float("inf")]


def _act(z, topo, alpha):
    with torch.no_grad():
        rms = float((z * z).mean().sqrt())
        if rms > 0:
            kink_margin[0] = min(kink_margin[0], float(z.abs().min()) / rms)
    a = topo.get("activation", "relu")
    if a == "relu":
        return F.relu(z)
    if a == "lrelu":
        return torch.maximum(topo.get("lrelu_alpha", 0.2) * z, z)
    return F.relu(z) + alpha * torch.clamp(z, max=0.0)


def forward(params, stats, topo, x, labels, train, dropout=None):
    """x[B,T,F] float64 tensor, labels int64[B].  Returns (loss, accuracy, new_stats, embedding0).
    dropout: None or {scope: (keep_mask tensor broadcastable to the layer output [B,T,C] / [B,C], keep_prob)} -- the
    tf.nn.dropout sites of class Model (models.py:70-72, 92-94: after BN of every layer but the last of its group), with
    the mask handed in (TF's random stream cannot be reproduced; the GPU build uses a counter-based mask)."""
    new_stats = {}
    kink_margin[0] = float("inf")

    def drop(h, scope):
        if not (train and dropout and scope in dropout):
            return h
        mask, keep = dropout[scope]
        return h * mask / keep

    def bn(r, scope, axes):
        g, b = params[scope + "/gamma:0"], params[scope + "/beta:0"]
        if train:
            mean = r.mean(dim=axes)
            var = ((r - mean) ** 2).mean(dim=axes)
            new_stats[scope + "/mean:0"] = stats[scope + "/mean:0"] * BN_DECAY + mean.detach() * (1 - BN_DECAY)
            new_stats[scope + "/variance:0"] = stats[scope + "/variance:0"] * BN_DECAY + var.detach() * (1 - BN_DECAY)
        else:
            mean, var = stats[scope + "/mean:0"], stats[scope + "/variance:0"]
        inv = g / torch.sqrt(var + BN_EPSILON)
        return r * inv + (b - mean * inv)

    h = x.transpose(1, 2)                                            # [B, C, T]
    for i, (K, d) in enumerate(zip(topo["kernel_sizes"], topo["dilations"])):
        sc = "frame_level_info_layer-%d" % i
        w = params[sc + "/w:0"].permute(2, 1, 0)                      # [Cout, Cin, K]
        z = F.conv1d(h, w, params[sc + "/b:0"], padding=(K - 1) * d // 2, dilation=d)
        alpha = params.get(sc + "/prelu/prelu:0")
        r = _act(z, topo, alpha.view(1, -1, 1) if alpha is not None else None)
        h = drop(bn(r.transpose(1, 2), sc, (0, 1)), sc).transpose(1, 2)
    ht = h.transpose(1, 2)                                           # [B, T, C]
    if topo.get("pooling", "stats") == "attention":                  # models.py:1036-1050, op for op
        h1, h2 = torch.chunk(ht, 2, dim=2)
        nl = torch.tanh(torch.einsum("ijk,kl->ijl", h1, params["attention/w:0"]) + params["attention/b:0"])
        att = torch.softmax(torch.einsum("ijk,k->ij", nl, params["attention/v:0"]), dim=-1)
        mu = torch.einsum("ijk,ij->ik", h2, att)
        var = torch.einsum("ijk,ij->ik", h2 * h2, att) - mu * mu
    else:
        mu = ht.mean(dim=1)
        var = ((ht - mu.unsqueeze(1)) ** 2).mean(dim=1)
    h = torch.cat([mu, torch.sqrt(var + VAR2STD_EPSILON)], dim=1)
    e0 = None
    for j in range(len(topo["embedding_sizes"])):
        sc = "embed_layer-%d" % j
        s = h @ params[sc + "/w:0"] + params[sc + "/b:0"]
        if j == 0:
            e0 = s
        alpha = params.get(sc + "/prelu/prelu:0")
        h = drop(bn(_act(s, topo, alpha), sc, (0,)), sc)
    head = topo.get("head")
    if head and head.get("type") == "am_softmax":        # build-defined head (not in the reference): Wang et al. 2018
        cos = F.normalize(h, dim=1, eps=1e-12) @ F.normalize(params["output/w:0"], dim=0, eps=1e-12)
        logits = head["scale"] * (cos - head["margin"] * F.one_hot(labels, cos.shape[1]).to(cos.dtype))
    else:
        logits = h @ params["output/w:0"] + params["output/b:0"]
    loss = F.cross_entropy(logits, labels, reduction="mean")
    beta = topo.get("l2_beta", 0.0)
    if beta:                                                           # models.py:811-842: tf.nn.l2_loss(t) = sum(t**2)/2
        l2 = 0.0
        am = bool(topo.get("head")) and topo["head"].get("type") == "am_softmax"
        for sc, coef in (("embed_layer-0", 0.1), ("embed_layer-1", 1.0), ("output", 1.0)):
            l2 = l2 + coef * 0.5 * (params[sc + "/w:0"] ** 2).sum()
            if not (am and sc == "output"):            # the build-defined AM head has no bias: output/b is not part of its model or penalty
                l2 = l2 + coef * 0.5 * (params[sc + "/b:0"] ** 2).sum()
        loss = loss + beta * l2
    acc = (logits.argmax(dim=1) == labels).double().mean()
    return loss, acc, new_stats, e0


def to_torch(weights, requires_grad_names=(), dtype=np.float64):
    out = {}
    for k, v in weights.items():
        t = torch.tensor(np.asarray(v, dtype=dtype))
        if k in requires_grad_names:
            t.requires_grad_(True)
        out[k] = t
    return out


def eval_batch(weights, topo, x, labels):
    p = to_torch(weights)
    with torch.no_grad():
        loss, acc, _, e0 = forward(p, p, topo, torch.tensor(np.asarray(x, np.float64)), torch.tensor(np.asarray(labels, np.int64)),
                                   train=False)
    return float(loss), float(acc), e0.numpy()


def train_step(weights, adam, topo, x, labels, lr, dropout=None, dtype=np.float64):
    """One optimizer step.  weights: {name: ndarray} (all variables incl. moving stats); adam: {"t": int, "m": {...},
    "v": {...}} (t = number of steps already taken).  Returns (loss, acc, new_weights, new_adam, grads).
    ``dtype=np.float32`` runs the same autograd in IEEE single precision: not an oracle, a yardstick -- how far ANY fp32 evaluation of
    the step lands from the float64 one (activation kinks, batch-norm backward over a few hundred rows); tests that follow a
    trajectory over several Adam steps size their bars with it."""
    names = trainable_names(topo)
    p = to_torch(weights, names, dtype)
    if dropout:
        dropout = {k: (torch.tensor(np.asarray(m, dtype)), float(keep)) for k, (m, keep) in dropout.items()}
    loss, acc, new_stats, _ = forward(p, p, topo, torch.tensor(np.asarray(x, dtype)),
                                      torch.tensor(np.asarray(labels, np.int64)), train=True, dropout=dropout)
    grads = torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True)      # AM-softmax head: output/b is unused
    grads = [g if g is not None else torch.zeros_like(p[n]) for g, n in zip(grads, names)]
    t = adam["t"] + 1
    lr_t = lr * np.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
    new_w = {k: np.array(v, dtype=dtype) for k, v in weights.items()}
    new_adam = {"t": t, "m": {}, "v": {}}
    gout = {}
    for n, g in zip(names, grads):
        g = g.numpy()
        gout[n] = g
        m = ADAM_B1 * adam["m"].get(n, 0.0) + (1 - ADAM_B1) * g
        v = ADAM_B2 * adam["v"].get(n, 0.0) + (1 - ADAM_B2) * g * g
        new_adam["m"][n], new_adam["v"][n] = m, v
        new_w[n] = new_w[n] - lr_t * m / (np.sqrt(v) + ADAM_EPS)
    for k, v in new_stats.items():
        new_w[k] = v.numpy()
    return float(loss.detach()), float(acc), new_w, new_adam, gout
