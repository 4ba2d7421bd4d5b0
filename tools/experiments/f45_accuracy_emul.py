"""Would Toom-Cook F(4, 5) (8 products per 4 output rows: 2 per row instead of F(2, 5)'s 3) keep the fp32tc arithmetic inside
1e-6?  Accuracy only: layer 1 of the fp32tc network is EMULATED with torch fp32 operations on the GPU (transforms as fp32
multiply-adds, the 8 products as fp32 GEMMs, taps transformed in fp64 and rounded once) inside the product's own pipeline, and the
x-vectors are compared with the fp64 oracle next to fp32 and the shipped fp32tc.   python tools/experiments/f45_accuracy_emul.py [n_seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fractions import Fraction as Fr                           # noqa: E402
from oracle import oracle                                      # noqa: E402  (the checker)
from toomcook_gen import INF, toomcook                         # noqa: E402
from xvector_amd import engine, hiplib, synthetic, topology   # noqa: E402

M, K = 4, 5
AT, G, BT = toomcook(M, K, [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2), INF])
N = M + K - 1
AT = np.array([[float(v) for v in r] for r in AT]); G = np.array([[float(v) for v in r] for r in G]); BT = np.array([[float(v) for v in r] for r in BT])
EMULATE = [False]
_pack, _layer = hiplib.pack_weights_toom, hiplib.tdnn_layer_toom


def pack(w3d):
    p = _pack(w3d)
    if p.K == K:
        p.U = torch.einsum("jk,kco->jco", torch.tensor(G, dtype=torch.float64, device=w3d.device), w3d.double()).float()
    return p


def layer(x, w, bias, scale, shift, act, alpha, row_valid, y, rows=None):
    if not (EMULATE[0] and w.K == K):
        return _layer(x, w, bias, scale, shift, act, alpha, row_valid, y, rows)
    assert act == 1
    R = x.shape[0] if rows is None else int(rows)
    Rp = (R + M - 1) // M * M
    xp = torch.zeros((Rp + N, w.cin), dtype=torch.float32, device=x.device)
    xp[2:2 + R] = x[:R, :w.cin]
    d = [xp[i:i + Rp:M] for i in range(N)]                    # d[i][p] = input row 4p + i - 2
    out = [None] * M
    for j in range(N):
        V = None
        for i in range(N):
            if BT[j, i] != 0:
                V = d[i] * float(BT[j, i]) if V is None else torch.add(V, d[i], alpha=float(BT[j, i]))
        P = V @ w.U[j]
        for q in range(M):
            if AT[q, j] != 0:
                out[q] = P * float(AT[q, j]) if out[q] is None else torch.add(out[q], P, alpha=float(AT[q, j]))
    o = torch.stack(out, 1).reshape(Rp, w.cout)[:R]
    t = torch.relu(o + bias if bias is not None else o)
    if scale is not None:
        t = t * scale
    if shift is not None:
        t = t + shift
    if row_valid is not None:
        t = t * (row_valid[:R] != 0).unsqueeze(1)
    y[:R, :w.cout] = t


hiplib.pack_weights_toom, hiplib.tdnn_layer_toom = pack, layer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
oracle.build()
topo = topology.get("ModelWithoutDropout")
print("%-13s %4s | %10s %10s %10s" % ("weights", "seed", "fp32", "fp32tc", "L1=F(4,5)"))
for kind in ("trained", "trained_like", "hostile"):
    for seed in range(200, 200 + (1 if kind == "trained" else n)):
        if kind == "trained":
            w, _ = synthetic.trained_checkpoint(topo, 23, n_spk=64, steps=300, seed=seed)
        else:
            w = getattr(synthetic, kind)(topo, 23, seed=seed)
        mats = synthetic.mfcc_like([30, 64, 150, 256, 400, 777], 23, seed=seed + 1)
        refs = [oracle.embed_utterance(m, w, topo, 25, 10000, np.float64) for m in mats]
        err = {}
        for name, prec, emu in (("fp32", "fp32", False), ("fp32tc", "fp32tc", False), ("f45", "fp32tc", True)):
            EMULATE[0] = emu
            got = engine.Extractor(engine.DeviceModel(w, topo, "cuda:0", precision=prec), 25, 10000, accuracy_probe=False).extract(mats)
            err[name] = max(oracle.rel_l2(g, r) for g, r in zip(got, refs))
        print("%-13s %4d | %10.2e %10.2e %10.2e" % (kind, seed, err["fp32"], err["fp32tc"], err["f45"]), flush=True)
