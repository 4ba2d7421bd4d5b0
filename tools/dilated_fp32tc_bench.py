"""The dilated class (ModelWithoutDropoutTdnn: kernels [5,3,3,1,1], dilations [1,2,3,1,1]; models.py:538-639) on BASELINE configs[1]'s
workload in the exact-fp32 arithmetics: "fp32" (direct K-tap kernels) against "fp32tc" (its K = 3 layers as F(2, 3) over every 2nd / 3rd
row, xv_tdnn_layer_toom_dilated_f32): utterances per second over 3 passes of 10 k utterances resident on the device, and the relative L2
of 16 x-vectors against the fp64 oracle.   python tools/dilated_fp32tc_bench.py [n_utts]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd")]
import torch
from oracle import oracle
from xvector_amd import engine, synthetic, topology
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
topo = topology.get("ModelWithoutDropoutTdnn")
w = synthetic.trained_like(topo, 23, seed=1)
lens = synthetic.utterance_lengths(n, 200, 400, 1234)
rng = np.random.default_rng(0)
pool = [(rng.standard_normal((400, 23)) * 3.0).astype(np.float32) for _ in range(64)]
mats = [pool[i % 64][:lens[i]] for i in range(n)]
refs = {i: oracle.embed_utterance(mats[i], w, topo, 25, 10000, np.float64) for i in range(0, n, max(1, n // 16))}
for precision in ("fp32", "fp32tc", "fp32", "fp32tc"):
    model = engine.DeviceModel(w, topo, "cuda:0", precision=precision)
    ex = engine.Extractor(model, 25, 10000)
    ex.extract(mats[:2000])
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vecs = ex.extract(mats)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    err = max(oracle.rel_l2(vecs[i], r) for i, r in refs.items())
    print("%-6s align %2d: %.1f utt/s (host packing and copies included; best of 3 passes over %d utterances), x-vector rel-L2 vs fp64 max %.2e; kernels: %s" % (
        precision, model.align, n / best, n, err, [type(L["wp"]).__name__ for L in model.layers]))
