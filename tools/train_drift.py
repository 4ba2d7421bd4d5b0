"""Step time of the training loop over a longer run, in chunks of 20 steps (development aid): is it the GPU warming up or the host?
python tools/train_drift.py [bf16x3|fp32] [chunks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import synthetic, topology as tp, trainer
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
topo = tp.get("ModelWithoutDropoutAMSoftmax")
w = synthetic.reference_init(topo, 23, 64, seed=1)
for k in list(w):
    if k.endswith("/w:0") and w[k].ndim == 3:
        w[k] = (w[k] * (np.sqrt(2.0 / (w[k].shape[0] * w[k].shape[1])) / 0.1)).astype(np.float32)
tr = trainer.Trainer(w, topo, "cuda:0", precision=prec)
rng = np.random.default_rng(0)
batches = [((rng.standard_normal((64, int(rng.integers(200, 401)), 23)) * 3).astype(np.float16), rng.integers(0, 64, 64).astype(np.int32)) for _ in range(20)]
for c in range(chunks):
    torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
    for b in batches:
        h0 = time.perf_counter()
        tr.step(b[0], b[1], 1e-3)
    torch.cuda.synchronize()
    print("chunk %2d: %.2f ms per step" % (c, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
