"""A longer training run (default 3000 lagged steps over all 201 minibatch lengths): step time per 250 steps, device memory held by the
caching allocator at the end of each block, final loss finite -- does anything creep (layout cache, workspaces, pinned buffers)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import synthetic, topology as tp, trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
topo = tp.get("ModelWithoutDropoutAMSoftmax")
w = synthetic.trained_like(topo, 23, num_classes=64, seed=1)
tr = trainer.Trainer(w, topo, "cuda:0", precision="bf16x3")
pool = list(synthetic.speaker_minibatches(256, 23, 64, 64, 200, 400, seed=5))
prev, t0, losses = None, time.perf_counter(), []
for i in range(steps):
    x, l = pool[i % len(pool)]
    h = tr.step_async(x, l, 1e-3)
    if prev is not None: losses.append(prev.result()[0])
    prev = h
    if (i + 1) % 250 == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        print("steps %5d..%5d: %.3f ms per step, reserved %.0f MB, allocated %.0f MB, loss %.4f" % (i - 248, i + 1, (now - t0) / 250 * 1e3,
              torch.cuda.memory_reserved() / 1e6, torch.cuda.memory_allocated() / 1e6, float(np.mean(losses[-250:]))), flush=True)
        t0 = time.perf_counter()
losses.append(prev.result()[0])
assert np.isfinite(losses).all()
print("layouts cached: %d, final loss %.4f" % (len(tr._layouts), losses[-1]))
