"""XCD-aware column-tile placement of the 256 x 256-tile f16bf8 GEMM (XV_TUNE_XCD_COLUMNS): the K = 5 / K = 7
layers of the default topology with XCDs 0-3 on column tile 0 and 4-7 on tile 1 (1) against every XCD on both tiles (0), interleaved
rounds in one process; the outputs must be bit-identical.  argv[1] = rows (default 262144)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for (cin, cout, K) in ((512, 512, 5), (512, 512, 7)):
    torch.manual_seed(K)
    w8 = hiplib.pack_weights_f16bf8(torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5)
    x = torch.relu(torch.randn((R, cin), device=dev)) * 1.3 - 0.4
    x8 = hiplib.SplitBuf(R, cin, dev, hiplib.FMT_SPLIT8); hiplib.split_encode(x, x8)
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    ys = {m: hiplib.SplitBuf(R, cout, dev, hiplib.FMT_SPLIT8) for m in (0, 1)}
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    times = {0: [], 1: []}
    for rnd in range(7):
        for m in (0, 1):
            hiplib.set_tuning(hiplib.TUNE_XCD_COLUMNS, m)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8): hiplib.tdnn_layer8(x8, R, w8, bias, None, None, 1, None, 1, rv, ys[m], status)
            b.record(); torch.cuda.synchronize()
            if rnd: times[m].append(a.elapsed_time(b) / 8)
    hiplib.set_tuning(hiplib.TUNE_XCD_COLUMNS, 0)
    same = torch.equal(ys[0].base, ys[1].base)
    med = {m: sorted(t)[len(t) // 2] for m, t in times.items()}
    print("K=%d: both tiles per XCD %.4f ms | one tile per XCD %.4f ms (%+.2f %%) | outputs bit-identical: %s" % (
        K, med[0], med[1], 100 * (med[1] / med[0] - 1), same))
