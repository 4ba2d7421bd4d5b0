"""Development aid: HBM rate of the self-attentive pooling kernels on one configs[1]-sized batch (262144 rows, A = 1536)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import engine, hiplib
dev = torch.device("cuda:0"); A = 1536
lens = np.random.default_rng(0).integers(200, 401, size=860)
lay = engine.BatchLayout(lens, 3); R = lay.rows
rs, rl = torch.from_numpy(lay.row_start).to(dev), torch.from_numpy(lay.row_len).to(dev)
u = torch.randn((R, A), device=dev); v = torch.randn(A, device=dev) * 0.04
h = torch.randn((R, 2 * A), device=dev); h2c = h[:, A:].contiguous()
scores = torch.empty(R, device=dev); att = torch.zeros(R, device=dev); out = torch.empty((len(lens), 2 * A), device=dev)
nl = torch.empty((R, A), device=dev); dp = torch.randn((len(lens), 2 * A), device=dev); dh = torch.zeros((R, 2 * A), device=dev)
datt = torch.zeros(R, device=dev); dsc = torch.zeros(R, device=dev); du = torch.empty((R, A), device=dev)
frames = int(lens.sum())


def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for name, fn, by in (
        ("attention_scores (read u)", lambda: hiplib.attention_scores(u, v, scores), 4 * A * R),
        ("attention_scores + tanh store (training)", lambda: hiplib.attention_scores(u, v, scores, nl), 8 * A * R),
        ("attention_softmax", lambda: hiplib.attention_softmax(scores, rs, rl, len(lens), att), 8 * frames),
        ("attention_pool, h2 contiguous", lambda: hiplib.attention_pool(h2c, att, rs, rl, len(lens), 400, 512, 1e-5, out), 4 * A * frames),
        ("attention_pool, h2 = right half of [R, 2A]", lambda: hiplib.attention_pool(h[:, A:], att, rs, rl, len(lens), 400, 512, 1e-5, out), 4 * A * frames),
        ("stats_pool (plain, same data)", lambda: hiplib.stats_pool(h2c, rs, rl, len(lens), 400, 512, 1e-5, out), 4 * A * frames),
        ("attention_pool_backward", lambda: hiplib.attention_pool_backward(h[:, A:], att, rs, rl, len(lens), 400, out, dp, dh[:, A:], datt), 8 * A * frames),
        ("attention_scores_backward", lambda: hiplib.attention_scores_backward(nl, dsc, v, du), 12 * A * R)):
    s = t(fn)
    print("%-46s %8.3f ms  %7.1f GB/s algorithmic (%.1f %% of 8 TB/s)" % (name, s * 1e3, by / s / 1e9, by / s / 8e12 * 100))
