"""The K = 5 / K = 7 layers of the default topology (and the dilated class's K = 3, d = 2 / 3 layers) in the direct exact-fp32 form (xv_tdnn_layer_f32, tdnn_gemm_dma_kernel) and in
the Toom-Cook F(2, K) form (xv_tdnn_layer_toom_f32): time per launch, algorithmic TF (2 K Cin Cout per row) against the 157.3 TF
fp32-MFMA peak, EXECUTED TF of the transformed form ((K + 1) / 2 products per row), and the relative L2 of both against a float64
matmul on a 2 k-row sample.  argv[1] = rows per batch (default 262144)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144


def timed(fn):
    ts = []
    for rnd in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4): fn()
        b.record(); torch.cuda.synchronize()
        if rnd: ts.append(a.elapsed_time(b) / 4)
    ts.sort()
    return ts[len(ts) // 2]


for (cin, cout, K, D) in ((512, 512, 5, 1), (512, 512, 7, 1), (512, 512, 3, 2), (512, 512, 3, 3), (512, 512, 3, 1)):
    torch.manual_seed(cin * 7 + cout + K)
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    wp = hiplib.pack_weights(w.reshape(K * cin, cout).contiguous())
    wt = hiplib.pack_weights_toom(w)
    x = torch.relu(torch.randn((R, cin), device=dev)) - 0.4
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    scale = torch.ones(cout, device=dev); shift = torch.zeros(cout, device=dev)
    yd = torch.empty((R, cout), device=dev); yt = torch.empty((R, cout), device=dev)
    fd = lambda: hiplib.tdnn_layer(x, wp, bias, scale, shift, 0, None, K, D, rv, yd)
    ft = lambda: hiplib.tdnn_layer(x, wt, bias, scale, shift, 0, None, K, D, rv, yt)
    md, mt = timed(fd), timed(ft)
    # float64 reference on rows [4096, 6144)
    lo, n, p = 4096, 2048, (K - 1) // 2 * D
    xs = x[lo - p:lo + n + p].double()
    ref = sum(xs[k * D:k * D + n] @ w[k].double() for k in range(K))
    ed = ((yd[lo:lo + n].double() - ref).norm() / ref.norm()).item()
    et = ((yt[lo:lo + n].double() - ref).norm() / ref.norm()).item()
    fl = 2.0 * R * cin * cout * K
    fx = 2.0 * R * cin * cout * (K + 1) / 2
    print("K=%d d=%d direct %.3f ms %.1f TF (%.3f of 157.3) err %.2e | toom %.3f ms: algorithmic %.1f TF (%.3f), executed %.1f TF (%.3f) err %.2e | x%.3f" % (
        K, D, md, fl / md / 1e9, fl / md / 1e9 / 157.3, ed, mt, fl / mt / 1e9, fl / mt / 1e9 / 157.3, fx / mt / 1e9, fx / mt / 1e9 / 157.3, et, md / mt))
