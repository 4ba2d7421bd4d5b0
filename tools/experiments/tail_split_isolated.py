"""Isolated launch time of the training forward GEMM (xv_tdnn_layer_bf16x3_moments, 512 -> 512) with and without the tail split, for
minibatch lengths on both sides of the slot boundary (64 chunks, 128-row tiles, 512 slots).  The tail split is NOT in the tree: the
kernel, launcher and binding changes are tools/experiments/tail_split.patch (git apply it, make, then run this); results and why it
was left out: profiles/r05_tail_split.txt, DESIGN 9.4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K in (5, 7):
    w = torch.randn((K, 512, 512), device=dev) / (K * 512) ** 0.5
    wp = hiplib.pack_weights_bf16x3(w)
    bias = torch.zeros(512, device=dev)
    for T in (200, 250, 253, 280, 300, 325, 350, 380, 400):
        R = 3 + 64 * (T + 3)
        x = torch.randn((R, 512), device=dev); y = torch.empty((R, 512), device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
        tiles = (R + 127) // 128 * 4
        res = []
        for on in (0, 1):
            hiplib.set_tuning(hiplib.TUNE_TAIL_SPLIT, on)
            ws = hiplib.train_gemm_workspace(R, 512, 512, dev)
            res.append(timeit(lambda: hiplib.tdnn_layer3_moments(x, R, wp, bias, 1, None, 1, rv, y, None, ws)))
        print("K=%d T=%d: %4d tiles (tail %3d): unsplit %6.1f us, tail split %6.1f us (%+.0f %%)" % (K, T, tiles, tiles % 512 if tiles > 512 else 0, res[0], res[1], (res[1] / res[0] - 1) * 100), flush=True)
