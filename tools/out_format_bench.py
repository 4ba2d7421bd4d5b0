import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch, numpy as np
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = 130889
for (cin, cout, K) in ((512, 512, 1), (512, 512, 5), (512, 1536, 1)):
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    wp = hiplib.pack_weights_bf16x3(w)
    x = torch.randn((R, cin), device=dev); xs = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, xs)
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    ys = hiplib.SplitBuf(R, cout, dev); yf = torch.empty((R, cout), device=dev)
    for name, y in (("split", ys), ("fp32 ", yf)):
        for _ in range(2): hiplib.tdnn_layer3(xs, R, wp, bias, None, None, 1, None, 1, rv, y)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): hiplib.tdnn_layer3(xs, R, wp, bias, None, None, 1, None, 1, rv, y)
        b.record(); torch.cuda.synchronize()
        print("cin %d cout %d K %d  out=%s: %.3f ms" % (cin, cout, K, name, a.elapsed_time(b) / 5))
