"""Where does a training step's wall time go beyond its kernels?  The lagged loop of bench.py on (a) lengths drawn per minibatch, (b) one
fixed length, with the host's enqueue time per step measured while the GPU is busy, and (c) the same steps enqueued with no read-back
at all (the host as far ahead as it can get: GPU-bound or enqueue-bound rate).   python tools/experiments/train_gap_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import synthetic, topology as tp, trainer
topo = tp.get("ModelWithoutDropoutAMSoftmax")
w = synthetic.trained_like(topo, 23, num_classes=64, seed=1)
tr = trainer.Trainer(w, topo, "cuda:0", precision=os.environ.get("PREC", "bf16x3"))
var = list(synthetic.speaker_minibatches(240, 23, 64, 64, 200, 400, seed=3))
fix = list(synthetic.speaker_minibatches(240, 23, 64, 64, 300, 300, seed=3))
for name, bs in (("T ~ U{200..400}", var), ("T = 300", fix), ("T ~ U{200..400} again", var)):
    for x, l in bs[:20]: tr.step(x, l, 1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prev, enq = None, 0.0
    for x, l in bs[20:]:
        a = time.perf_counter(); h = tr.step_async(x, l, 1e-3); enq += time.perf_counter() - a
        if prev is not None: prev.result()
        prev = h
    prev.result(); torch.cuda.synchronize()
    n = len(bs) - 20
    print("%-22s lagged loop %.3f ms/step, host enqueue %.3f ms/step" % (name, (time.perf_counter() - t0) / n * 1e3, enq / n * 1e3), flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for x, l in bs[20:120]: tr.step_async(x, l, 1e-3)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-22s no read-back: enqueue %.3f ms/step, until the GPU is done %.3f ms/step" % (name, (t1 - t0) / 100 * 1e3, (t2 - t0) / 100 * 1e3), flush=True)
