"""Soak: the bf16x3 training step with its K = 1 layers on split-format copies (default) against the same step on fp32 rows
(XVECTOR_TRAIN_SPLIT_K1=0), random topologies -- losses and every gradient tensor must agree BIT FOR BIT.   python tools/experiments/split_k1_soak.py [seed0 seed1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np
from xvector_amd import synthetic, topology, trainer
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 30)
bad = used = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    topo = topology.get("ModelWithoutDropout" if seed % 3 else "ModelL2LossWithoutDropoutLRelu")
    w32 = lambda: int(rng.choice([32, 64, 96, 128, 40, 72]))
    topo["layer_sizes"] = [w32() for _ in range(5)]
    topo["kernel_sizes"] = [5] + [int(rng.choice([1, 1, 3, 5])) for _ in range(4)]
    topo["dilations"] = [1] * 5
    topo["embedding_sizes"] = [32, 32]
    w = synthetic.trained_like(topo, 23, num_classes=10, seed=seed)
    B, T = int(rng.integers(2, 12)), int(rng.integers(20, 260))
    x = (rng.standard_normal((B, T, 23)) * 3).astype(np.float32); lab = rng.integers(0, 10, B)
    res = []
    for flag in ("1", "0"):
        os.environ["XVECTOR_TRAIN_SPLIT_K1"] = flag
        tr = trainer.Trainer(w, topo, precision="bf16x3")
        loss, acc, grads = tr.gradients(x, lab)
        res.append((loss, {n: g.cpu().numpy().copy() for n, g in grads.items()}, bool(tr._splits)))
    used += res[0][2]
    same = res[0][0] == res[1][0] and all(np.array_equal(res[0][1][n], res[1][1][n]) for n in res[0][1])
    if not same:
        bad += 1
        print("seed", seed, "DIFFERS", topo["layer_sizes"], topo["kernel_sizes"])
print("seeds %d..%d: %d took the split path, %d differ" % (lo, hi - 1, used, bad))
sys.exit(1 if bad else 0)
