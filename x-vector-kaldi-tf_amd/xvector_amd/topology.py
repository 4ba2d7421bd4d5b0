"""Topologies of the reference's model classes that share the extraction graph.

Each entry restates the constants of one ``build_model`` in the reference's
``local/tf/models.py`` (line numbers below); all of them share ``Model``'s load / extract code, so
on the extraction path a class is fully described by these numbers.
"""
import copy

BN_EPSILON = 1e-3          # local/tf/tf_block.py:9   batch_norm_wrapper(epsilon=1e-3)
VAR2STD_EPSILON = 1e-5     # local/tf/models.py:16

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_PRELU = 0, 1, 2, 3
ACT_CODES = {"none": ACT_NONE, "relu": ACT_RELU, "lrelu": ACT_LRELU, "prelu": ACT_PRELU}

_BASE = dict(
    layer_sizes=[512, 512, 512, 512, 1536],     # models.py:27
    kernel_sizes=[5, 5, 7, 1, 1],               # models.py:28
    dilations=[1, 1, 1, 1, 1],
    embedding_sizes=[512, 512],                 # models.py:29
    activation="relu",
    lrelu_alpha=0.2,                            # models.py:912
    l2_beta=0.0,                                # training only: loss += l2_beta*(0.1*l2(embed-0) + l2(embed-1) + l2(output))
    dropout=False,                              # training only: tf.nn.dropout sites after BN (class Model, models.py:70-72,92-94)
    head=None,                                  # training only: None = softmax-CE on output/xw_plus_b (models.py:96-113)
    init="default",                             # build_model only: "default" = truncated_normal(0.1) / b = 0.1 / Xavier output
                                                # (models.py:56-58,98-100); "he" = models.py:1158-1163,1181-1185,1205-1210
    pooling="stats",                            # "stats": tf.nn.moments over time (models.py:75-76); "attention": see below
)

TOPOLOGIES = {
    # class name in the reference           : constants
    "Model":                                 dict(_BASE, dropout=True),                      # models.py:20-128
    "ModelWithoutDropout":                   dict(_BASE),                                    # models.py:436-534
    "ModelWithoutDropoutTdnn":               dict(_BASE, kernel_sizes=[5, 3, 3, 1, 1],       # models.py:538-639
                                                  dilations=[1, 2, 3, 1, 1]),
    "ModelWithoutDropoutPRelu":              dict(_BASE, activation="prelu"),                # models.py:643-742
    "ModelL2LossWithoutDropoutPRelu":        dict(_BASE, activation="prelu", l2_beta=0.0002),   # models.py:746-862 (beta :756)
    "ModelL2LossWithoutDropoutLRelu":        dict(_BASE, activation="lrelu", l2_beta=0.0002),   # models.py:866-981 (beta :876)
    "ModelL2LossWithoutDropoutReluHeInit":   dict(_BASE, l2_beta=0.0002, init="he"),            # models.py:1118-1244 (beta :1128)
    # self-attentive pooling: the last layer is 6*512 wide and split into h1 | h2; weights softmax_t(v . tanh(h1 W + b)) pool h2
    # (models.py:985-1114: sizes :992, split/attention :1036-1050, beta :995); variables attention/{w,b,v}:0
    "ModelL2LossWithoutDropoutLReluAttention": dict(_BASE, layer_sizes=[512, 512, 512, 512, 3072], activation="lrelu",
                                                    l2_beta=0.0002, pooling="attention"),
    # BUILD-DEFINED (not in the reference; BASELINE configs[4] asks for an AM-softmax head): ModelWithoutDropout's network
    # with logits = scale*(cos(x, w_j) - margin*[j == y]) on the L2-normalised embed_layer-1 output and output/w columns
    "ModelWithoutDropoutAMSoftmax":          dict(_BASE, head=dict(type="am_softmax", scale=30.0, margin=0.2)),
}


def get(name):
    if name not in TOPOLOGIES:
        raise KeyError("unknown model class '%s'" % name)
    return copy.deepcopy(TOPOLOGIES[name])


def is_attention(topo):
    return topo.get("pooling", "stats") == "attention"


def pooled_dim(topo):
    """Width of the pooled vector: [mean | std] of all channels, or of the h2 half with attention (models.py:1038,1052)."""
    c = topo["layer_sizes"][-1]
    return c if is_attention(topo) else 2 * c


def max_halo(topo):
    """Largest one-sided SAME padding over the frame-level layers: (K-1)*d/2 (all K odd)."""
    return max((k - 1) * d // 2 for k, d in zip(topo["kernel_sizes"], topo["dilations"]))


def flops_per_frame(topo, feat_dim):
    """Algorithmic FLOPs/frame of the frame-level layers (2*MAC, full taps at padded edges)."""
    f, prev = 0, feat_dim
    for k, c in zip(topo["kernel_sizes"], topo["layer_sizes"]):
        f += 2 * k * prev * c
        prev = c
    if is_attention(topo):
        f += 2 * (prev // 2) ** 2                  # h1 . W  (models.py:1045)
    return f


def flops_per_utt(topo, embedding_index=0):
    """Algorithmic FLOPs of the segment-level part for one chunk (embed-0, optionally embed-1)."""
    pooled = pooled_dim(topo)
    f = 2 * pooled * topo["embedding_sizes"][0]
    if embedding_index == 1:
        f += 2 * topo["embedding_sizes"][0] * topo["embedding_sizes"][1]
    return f
