"""The role-split f16bf8 kernel (XV_F16BF8_RS=1, 256 x 128 tiles: fp16 main terms on 16x16x32 MFMAs in one wave of a SIMD, the scaled
cross terms in the other) against the shipped 256 x 256 kernel: results (relative L2 on the decoded output) and time.  Run once with
and once without the variable; this script prints a checksum, the error against an fp32 torch matmul on a slice, and the time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
torch.manual_seed(0)
for (cin, cout, K) in ((512, 512, 5), (512, 512, 7)):
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    w8 = hiplib.pack_weights_f16bf8(w)
    x = torch.relu(torch.randn((R, cin), device=dev)) * 1.3 - 0.4
    x8 = hiplib.SplitBuf(R, cin, dev, hiplib.FMT_SPLIT8); hiplib.split_encode(x, x8)
    bias = torch.randn(cout, device=dev) * 0.1; rv = torch.ones(R, dtype=torch.uint8, device=dev)
    rv[1000:1003] = 0
    y8 = hiplib.SplitBuf(R, cout, dev, hiplib.FMT_SPLIT8); status = torch.zeros(1, dtype=torch.int32, device=dev)
    hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, 512)
    fn = lambda: hiplib.tdnn_layer8(x8, R, w8, bias, None, None, 1, None, 1, rv, y8, status)
    fn(); torch.cuda.synchronize()
    got = hiplib.split_decode(y8, R)
    # reference on rows 2000..2512 (interior): exact fp32 conv via matmul over taps
    r0, n = 2000, 512
    half = (K - 1) // 2
    ref = torch.zeros((n, cout), device=dev, dtype=torch.float64)
    for t in range(K):
        ref += x[r0 - half + t:r0 - half + t + n].double() @ w[t].double()
    ref = torch.relu(ref + bias.double())
    err = float((got[r0:r0 + n].double() - ref).norm() / ref.norm())
    ts = []
    for rnd in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8): fn()
        b.record(); torch.cuda.synchronize()
        if rnd: ts.append(a.elapsed_time(b) / 8)
    ts.sort()
    hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, 0)
    print("K %d: rel-L2 vs fp64 on 512 rows %.3e  gap rows zero %s  checksum %.6f  status %d  median %.3f ms (min %.3f)" % (
        K, err, bool((got[1000:1003] == 0).all()), float(got.double().abs().mean()), int(status.item()), ts[len(ts) // 2], ts[0]))
