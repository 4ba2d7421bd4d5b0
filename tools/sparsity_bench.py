"""Does operand sparsity buy clock on the power-limited GEMM?  K=7 layer on dense vs 50 %-zero (ReLU-like) activations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = 130889; cin = cout = 512; K = 7
w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
wp = hiplib.pack_weights_bf16x3(w)
bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
ys = hiplib.SplitBuf(R, cout, dev)
for name, x in (("dense N(0,1)", torch.randn((R, cin), device=dev)),
                ("relu(N(0,1)) (50% zeros)", torch.relu(torch.randn((R, cin), device=dev))),
                ("relu(N(0,1))*s+t (BN-affine, dense)", torch.relu(torch.randn((R, cin), device=dev)) * 1.3 - 0.4),
                ("75% zeros", torch.randn((R, cin), device=dev) * (torch.rand((R, cin), device=dev) > 0.75)),
                ("all zeros", torch.zeros((R, cin), device=dev))):
    xs = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, xs)
    for _ in range(3): hiplib.tdnn_layer3(xs, R, wp, bias, None, None, 1, None, 1, rv, ys)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): hiplib.tdnn_layer3(xs, R, wp, bias, None, None, 1, None, 1, rv, ys)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%-40s %.3f ms  (%.1f%% of 2.5 PF)" % (name, ms, 6.0 * R * cin * cout * K / ms / 1e9 / 25))
