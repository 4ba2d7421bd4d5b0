"""Extractor throughput (host matrices -> host x-vectors) of the default topology at another feature dimension (argv[1], default 30: the
VoxCeleb recipes' MFCCs), f16bf8 against bf16x3: with K = 5 a 30-dimensional first layer does not fit the first-layer kernel, and the
f16bf8 model runs it on the general bf16x3 GEMM plus one encoding pass."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/x-vector-kaldi-tf_amd")
import numpy as np, torch
from xvector_amd import engine, synthetic, topology as tp
feat = int(sys.argv[1]) if len(sys.argv) > 1 else 30
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, feat, seed=1)
utts = synthetic.make_utterances(10000, 200, 400, feat, 1234)
mats = [m for _, m in utts]
for prec in ("f16bf8", "bf16x3"):
    model = engine.DeviceModel(w, topo, "cuda:0", precision=prec)
    ex = engine.Extractor(model, 25, 10000, accuracy_probe=False)
    ex.extract(mats[:2000]); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.time(); v = ex.extract(mats); torch.cuda.synchronize(); best = min(best, time.time() - t0)
    print("feat %d %s (f16bf8=%s, first kernel=%s): host->host %.0f utt/s" % (feat, prec, model.f16bf8, model.first is not None, len(mats) / best))
