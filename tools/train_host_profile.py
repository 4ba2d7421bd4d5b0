import cProfile, pstats, sys, os, time
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import synthetic, topology as tp, trainer
topo = tp.get("ModelWithoutDropoutAMSoftmax")
w = synthetic.reference_init(topo, 23, 64, seed=1)
for k in list(w):
    if k.endswith("/w:0") and w[k].ndim == 3:
        w[k] = (w[k] * (np.sqrt(2.0 / (w[k].shape[0] * w[k].shape[1])) / 0.1)).astype(np.float32)
tr = trainer.Trainer(w, topo, "cuda:0", precision="bf16x3")
rng = np.random.default_rng(0)
batches = [((rng.standard_normal((64, int(rng.integers(200, 401)), 23)) * 3).astype(np.float16), rng.integers(0, 64, 64).astype(np.int32)) for _ in range(24)]
for b in batches[:4]: tr.step(b[0], b[1], 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for b in batches[4:]: tr.step(b[0], b[1], 1e-3)
pr.disable()
torch.cuda.synchronize()
print("20 steps: %.2f ms per step (under cProfile)" % ((time.perf_counter() - t0) / 20 * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
