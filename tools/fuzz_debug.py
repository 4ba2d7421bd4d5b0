"""Layer-by-layer comparison of the f16bf8 path against the fp64 oracle for one topology (development aid)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from oracle import oracle
from xvector_amd import engine, hiplib, synthetic, topology as tp
F = 23
topo = dict(layer_sizes=[512, 96, 40, 32, 48], kernel_sizes=[5, 1, 1, 7, 3], dilations=[1, 1, 1, 1, 1], embedding_sizes=[16, 16],
            activation="lrelu", lrelu_alpha=0.2, pooling="stats")
w = synthetic.trained_like(topo, F, 8, seed=5)
rng = np.random.default_rng(0)
lens = [int(a) for a in (sys.argv[1:] or [300, 34, 200])]
mats = [(rng.standard_normal((t, F)) * 3).astype(np.float32) for t in lens]
model = engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8")
dev = model.device
layout = engine.BatchLayout(lens, model.gap, model.align)
host = np.zeros((layout.rows, model.in_dim), np.float32)
layout.pack(mats, host)
x = torch.from_numpy(host).to(dev); rv = torch.from_numpy(layout.row_valid()).to(dev)
R = layout.rows
S8 = hiplib.FMT_SPLIT8
status = torch.zeros(1, dtype=torch.int32, device=dev)
L = model.layers[0]
h = hiplib.SplitBuf(R, L["cout"], dev, S8)
hiplib.tdnn_first(x, R, model.first, L["bias"], L["scale"], L["shift"], model.act, L["alpha"], L["dil"], rv, h, status)
# oracle intermediates per utterance
refs = [oracle.forward(m, w, topo, np.float64, 0, True) for m in mats]
def check(i, buf):
    got = hiplib.split_decode(buf, R).cpu().numpy() if isinstance(buf, hiplib.SplitBuf) else buf.cpu().numpy()
    worst = 0
    for u, (s, n) in enumerate(zip(layout.row_start, layout.row_len)):
        ref = refs[u][1][i] if isinstance(refs[u], tuple) else None
        worst = max(worst, oracle.rel_l2(got[s:s + n], ref))
    print("layer %d (K=%d, %d -> %d): worst rel L2 %.3e  finite %s" % (i, model.layers[i]["K"], model.layers[i]["cin"], model.layers[i]["cout"], worst, np.isfinite(got).all()))
check(0, h)
for i in range(1, len(model.layers)):
    L = model.layers[i]
    y = hiplib.SplitBuf(R, L["cout"], dev, S8) if i < len(model.layers) - 1 else torch.empty((R, L["cout"]), dtype=torch.float32, device=dev)
    hiplib.tdnn_layer8(h, R, L["wp8"], L["bias"], L["scale"], L["shift"], model.act, L["alpha"], L["dil"], rv, y, status)
    check(i, y)
    h = y
print("status", int(status.item()))

# the engine path on the same model: batch-size dependence
lens2 = [789, 820, 860, 660, 675, 34, 592]
mats2 = [(rng.standard_normal((t, F)) * 3).astype(np.float32) for t in lens2]
refs2 = [oracle.embed_utterance(m, w, topo, 1, -1, np.float64) for m in mats2]
for mbr in (262144, 700, 2000):
    for rep in range(2):
        ex = engine.Extractor(model, 1, -1, max_batch_rows=mbr)
        got = ex.extract(mats2)
        print("engine mbr %d rep %d:" % (mbr, rep), ["%.1e" % oracle.rel_l2(g, r) for g, r in zip(got, refs2)], ex.stats.get("fallback_windows", 0))
