"""Per-layer timing of the bf16x3 GEMM on random data (development aid).  XVECTOR_HIP_LIB selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = 130889
print("lib:", hiplib.SO_PATH)
for (cin, cout, K) in ((512, 512, 1), (512, 1536, 1), (512, 512, 5), (512, 512, 7)):
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    wp = hiplib.pack_weights_bf16x3(w)
    x = torch.randn((R, cin), device=dev); xs = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, xs)
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    ys = hiplib.SplitBuf(R, cout, dev)
    blk = torch.empty(hiplib.block_stats_floats(R, cout), device=dev)
    outs = [("split", lambda: hiplib.tdnn_layer3(xs, R, wp, bias, None, None, 1, None, 1, rv, ys))]
    if cout == 1536:
        outs.append(("pool ", lambda: hiplib.tdnn_layer_pool(xs, R, wp, bias, None, None, 1, None, 1, rv, blk)))
    for name, fn in outs:
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print("cin %d cout %4d K %d out=%s: %.3f ms  %.0f TF executed (%.1f%% of 2.5 PF)" %
              (cin, cout, K, name, ms, 6.0 * R * cin * cout * K / ms / 1e9, 6.0 * R * cin * cout * K / ms / 1e9 / 25))
