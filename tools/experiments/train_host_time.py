import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/x-vector-kaldi-tf_amd") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import synthetic, topology as tp, trainer
topo = tp.get("ModelWithoutDropoutAMSoftmax")
w = synthetic.trained_like(topo, 23, num_classes=64, seed=1)
tr = trainer.Trainer(w, topo, "cuda:0", precision="bf16x3")
bs = list(synthetic.speaker_minibatches(40, 23, 64, 64, 200, 400, seed=3))
for x, l in bs[:5]: tr.step(x, l, 1e-3)
torch.cuda.synchronize()
enq, tot = [], []
for x, l in bs[5:]:
    torch.cuda.synchronize()
    t0 = time.perf_counter(); h = tr.step_async(x, l, 1e-3); t1 = time.perf_counter(); h.result(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print("enqueue ms: mean %.3f min %.3f | enqueue + wait: mean %.3f" % (np.mean(enq) * 1e3, np.min(enq) * 1e3, np.mean(tot) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for x, l in bs[5:25]: tr.step_async(x, l, 1e-3)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
for mode in ("sync", "lagged", "sync", "lagged"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prev = None
    for x, l in bs[5:]:
        if mode == "sync":
            tr.step(x, l, 1e-3)
        else:
            h = tr.step_async(x, l, 1e-3)
            if prev is not None: prev.result()
            prev = h
    if prev is not None: prev.result()
    torch.cuda.synchronize()
    print(mode, "ms per step %.3f" % ((time.perf_counter() - t0) / len(bs[5:]) * 1e3))
