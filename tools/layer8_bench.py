"""bf16x3 against f16bf8 (one fp16 MFMA + one scaled bf8 MFMA per product), layer by layer, interleaved rounds in ONE process.
argv[1] = rows per batch (default 262144)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ROUNDS, REPS = 5, 8
print("lib:", hiplib.SO_PATH, "rows:", R)
for (cin, cout, K) in ((512, 512, 5), (512, 512, 7), (512, 512, 1), (512, 1536, 1)):
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    w3, w8 = hiplib.pack_weights_bf16x3(w), hiplib.pack_weights_f16bf8(w)
    x = torch.relu(torch.randn((R, cin), device=dev)) * 1.3 - 0.4
    x3 = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, x3)
    x8 = hiplib.SplitBuf(R, cin, dev, hiplib.FMT_SPLIT8); hiplib.split_encode(x, x8)
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    y3 = hiplib.SplitBuf(R, cout, dev); y8 = hiplib.SplitBuf(R, cout, dev, hiplib.FMT_SPLIT8)
    blk = torch.empty(hiplib.block_stats_floats(R, cout), device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    if cout == 1536:
        fns = {"bf16x3": lambda: hiplib.tdnn_layer_pool(x3, R, w3, bias, None, None, 1, None, 1, rv, blk),
               "f16bf8": lambda: hiplib.tdnn_layer_pool8(x8, R, w8, bias, None, None, 1, None, 1, rv, blk)}
    else:
        fns = {"bf16x3": lambda: hiplib.tdnn_layer3(x3, R, w3, bias, None, None, 1, None, 1, rv, y3),
               "f16bf8": lambda: hiplib.tdnn_layer8(x8, R, w8, bias, None, None, 1, None, 1, rv, y8, status),
               "f16bf8 -> bf16 split": lambda: hiplib.tdnn_layer8(x8, R, w8, bias, None, None, 1, None, 1, rv, y3, status)}
    times = {(n, rows): [] for n in fns for rows in (128, 256, 512)}
    for rnd in range(ROUNDS + 1):
        for rows in (128, 256, 512):
            hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, rows)
            for n, fn in fns.items():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(REPS): fn()
                b.record(); torch.cuda.synchronize()
                if rnd: times[(n, rows)].append(a.elapsed_time(b) / REPS)
    hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, 0)
    # accuracy of both against the fp32 matmul of torch on a slice (rough; the parity tests are the real check)
    for (n, rows), t in times.items():
        t = sorted(t); med = t[len(t) // 2]
        alg = 2.0 * R * cin * cout * K / 1e9
        print("cin %d cout %4d K %d %-22s tile %3d: median %.3f ms (min %.3f)  %.0f TF algorithmic" % (cin, cout, K, n, rows, med, t[0], alg / med))
    assert int(status.item()) == 0
