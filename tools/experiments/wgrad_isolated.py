"""Isolated launch times of the training step's three GEMM families on the minibatch shapes (64 chunks x 300 frames + gaps):
forward (bf16x3), input gradient (bf16x3 on flipped weights), weight gradient (xv_wgrad_bf16x3 + its split merge), per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0")
R = 3 + 64 * 303


def timeit(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("R = %d rows" % R)
for cin, cout, K in ((24, 512, 5), (512, 512, 5), (512, 512, 7), (512, 512, 1), (512, 1536, 1)):
    x = torch.randn((R, cin), device=dev); dz = torch.randn((R, cout), device=dev)
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    wf = hiplib.pack_weights_bf16x3(w)
    wt = hiplib.pack_weights_bf16x3(w.flip(0).transpose(1, 2).contiguous())
    y = torch.empty((R, cout), device=dev); dx = torch.empty((R, cin), device=dev); dw = torch.empty((K, cin, cout), device=dev)
    rv = torch.ones(R, dtype=torch.uint8, device=dev)
    fl = 2.0 * R * cin * cout * K
    tf = timeit(lambda: hiplib.tdnn_layer3(x, R, wf, None, None, None, 1, None, 1, rv, y))
    td = timeit(lambda: hiplib.tdnn_layer3(dz, R, wt, None, None, None, 0, None, 1, rv, dx)) if cin % 8 == 0 and cin > 24 else float("nan")
    tw = timeit(lambda: hiplib.wgrad(x, dz, K, 1, dw, "bf16x3"))
    print("%4d -> %4d K=%d: forward %6.1f us (%5.0f TF)  dgrad %6.1f us  wgrad %6.1f us (%5.0f TF)" % (cin, cout, K, tf, fl / tf / 1e6, td, tw, fl / tw / 1e6))
