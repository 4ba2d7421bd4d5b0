"""The Toom-Cook arithmetic on checkpoints nothing forbids (synthetic.hostile: Student-t weights, channel scales over three decades,
near-dead channels) next to trained-like draws and a trained checkpoint: worst relative L2 of six MFCC-like utterances against the
fp64 oracle, forced fp32 (direct K-tap form) and forced fp32tc, per checkpoint.   python tools/fp32tc_hostile_sweep.py [n_seeds [model class]]
(model class: ModelWithoutDropout (default) or e.g. ModelWithoutDropoutTdnn, whose K = 3 layers run as F(2, 3) on dilation-strided rows)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
from oracle import oracle                                      # noqa: E402  (the checker)
from xvector_amd import engine, synthetic, topology           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
oracle.build()
topo = topology.get(sys.argv[2] if len(sys.argv) > 2 else "ModelWithoutDropout")
print("model class:", sys.argv[2] if len(sys.argv) > 2 else "ModelWithoutDropout")
print("%-13s %4s | %10s %10s | ratio" % ("weights", "seed", "fp32", "fp32tc"))
worst_ratio, worst_tc = 0.0, 0.0
for kind in ("trained", "trained_like", "hostile"):
    for seed in range(200, 200 + (2 if kind == "trained" else n)):
        if kind == "trained":
            w, _ = synthetic.trained_checkpoint(topo, 23, n_spk=64, steps=300, seed=seed)
        else:
            w = getattr(synthetic, kind)(topo, 23, seed=seed)
        mats = synthetic.mfcc_like([30, 64, 150, 256, 400, 777], 23, seed=seed + 1)
        refs = [oracle.embed_utterance(m, w, topo, 25, 10000, np.float64) for m in mats]
        err = {}
        for prec in ("fp32", "fp32tc"):
            got = engine.Extractor(engine.DeviceModel(w, topo, "cuda:0", precision=prec), 25, 10000, accuracy_probe=False).extract(mats)
            err[prec] = max(oracle.rel_l2(g, r) for g, r in zip(got, refs))
        worst_ratio = max(worst_ratio, err["fp32tc"] / err["fp32"]); worst_tc = max(worst_tc, err["fp32tc"])
        print("%-13s %4d | %10.2e %10.2e | %.2f" % (kind, seed, err["fp32"], err["fp32tc"], err["fp32tc"] / err["fp32"]), flush=True)
print("worst fp32tc error %.2e, worst fp32tc / fp32 ratio %.2f" % (worst_tc, worst_ratio))
