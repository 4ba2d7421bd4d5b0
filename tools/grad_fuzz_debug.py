"""One case of tests/test_gpu_fuzz.py::test_random_topologies_gradients_match_autograd, every gradient tensor printed (norm of ours, of the
float64 autograd oracle's, relative L2).  argv: seed case."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "x-vector-kaldi-tf_amd"))
from oracle import train_ref
from xvector_amd import synthetic, trainer
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(5):
    width = lambda lo, hi: int(rng.integers(lo, hi)) // 4 * 4                        # noqa: E731
    attention = rng.random() < 0.4
    ks = [int(rng.choice([1, 3, 5, 7])) for _ in range(5)]
    ds = [int(rng.choice([1, 2])) if 1 < k < 7 else 1 for k in ks]
    topo = dict(layer_sizes=[width(16, 100), width(16, 100), width(16, 100), width(16, 100), width(16, 120) // 8 * 8],
                kernel_sizes=ks, dilations=ds, embedding_sizes=[width(8, 48), width(8, 48)],
                activation=str(rng.choice(["relu", "lrelu", "prelu"])), lrelu_alpha=0.2, l2_beta=float(rng.choice([0.0, 0.0002])),
                dropout=False, head=None, pooling="attention" if attention else "stats")
    F, classes, B, T = int(rng.choice([23, 24, 13])), 7, int(rng.integers(3, 9)), int(rng.integers(40, 230))
    w = synthetic.trained_like(topo, F, classes, seed=int(rng.integers(1 << 30)))
    x = (rng.standard_normal((B, T, F)) * 3).astype(np.float32)
    lab = rng.integers(0, classes, B)
    if case != want:
        continue
    print(topo, "F", F, "B", B, "T", T)
    loss, acc, grads = trainer.Trainer(w, topo).gradients(x, lab)
    rl, ra, _, _, rg = train_ref.train_step(w, {"t": 0, "m": {}, "v": {}}, topo, x, lab, 1e-3)
    print("loss", loss, rl)
    for n, ref in rg.items():
        g = grads[n].cpu().numpy().astype(np.float64)
        print("%-40s ours %.4e  ref %.4e  rel %.3e" % (n, np.linalg.norm(g), np.linalg.norm(ref), np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 1e-30)))
