"""The first-layer kernel (K=5, 23 -> 512, split8 output).  FIRST_BENCH_ZERO=1: zero frames and weights (full clock)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ZERO = bool(os.environ.get("FIRST_BENCH_ZERO", ""))
K, cin, cout = 5, 23, 512
w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
x = torch.zeros((R, 24), device=dev); x[:, :cin] = torch.randn((R, cin), device=dev) * 3
if ZERO: w.zero_(); x.zero_()
pw = hiplib.pack_first_bf16x3(w)
bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
for fmt, name in ((hiplib.FMT_SPLIT8, "split8"), (hiplib.FMT_SPLIT, "bf16 split")):
    y = hiplib.SplitBuf(R, cout, dev, fmt)
    fn = lambda: hiplib.tdnn_first(x, R, pw, bias, None, None, 1, None, 1, rv, y, status if fmt == hiplib.FMT_SPLIT8 else None)
    ts = []
    for rnd in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8): fn()
        b.record(); torch.cuda.synchronize()
        if rnd: ts.append(a.elapsed_time(b) / 8)
    ts.sort(); med = ts[len(ts) // 2]
    print("first layer -> %-10s %s: median %.3f ms (min %.3f)  output %.2f TB/s" % (name, "zeros" if ZERO else "random", med, ts[0], R * cout * 4 / med / 1e9))
