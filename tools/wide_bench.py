"""The 256 x 256-tile f16bf8 kernels (32 x 32 and 16 x 16 MFMA shapes) on layers 1 / 2 (K = 5, 7; 512 -> 512).  WIDE_BENCH_ZERO=1 runs it on zero weights and
frames: the chip then holds its full clock, and the difference to the random-data time is the power limit, the rest structure.
argv[1] = rows (default 262144)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ZERO = bool(os.environ.get("WIDE_BENCH_ZERO", ""))
SPARSE = bool(os.environ.get("WIDE_BENCH_SPARSE", ""))    # frames = relu(randn): half of them exactly zero
cout = 512
for cin, K in [(int(c), int(k)) for c, k in (a.split(':') for a in os.environ.get('WIDE_BENCH_SHAPES', '512:5,512:7').split(','))]:
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    x = torch.relu(torch.randn((R, cin), device=dev)) * 1.3 - 0.4
    if SPARSE: x = torch.relu(torch.randn((R, cin), device=dev))
    if ZERO: w.zero_(); x.zero_()
    w8 = hiplib.pack_weights_f16bf8(w)
    x8 = hiplib.SplitBuf(R, cin, dev, hiplib.FMT_SPLIT8); hiplib.split_encode(x, x8)
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    y8 = hiplib.SplitBuf(R, cout, dev, hiplib.FMT_SPLIT8); status = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lambda: hiplib.tdnn_layer8(x8, R, w8, bias, None, None, 1, None, 1, rv, y8, status)
    # the 32 x 32 MFMA form (512) and the 16 x 16 form (1024) of the 256 x 256 tile, alternating rounds in one process
    ts = {512: [], 1024: []}
    outs = {}
    for rnd in range(7):
        for rows in (512, 1024):
            hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, rows)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8): fn()
            b.record(); torch.cuda.synchronize()
            if rnd: ts[rows].append(a.elapsed_time(b) / 8)
            else: outs[rows] = hiplib.split_decode(y8, R).clone()
    hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, 0)
    d = (outs[512] - outs[1024]).double().norm() / outs[512].double().norm()
    for rows in (512, 1024):
        t = sorted(ts[rows]); med = t[len(t) // 2]
        print("cin %d K %d %s %s: median %.3f ms (min %.3f)  %.0f TF algorithmic" % (cin, K, "zeros" if ZERO else "sparse" if SPARSE else "random",
              "32x32" if rows == 512 else "16x16", med, t[0], 2.0 * R * cin * cout * K / 1e9 / med))
    print("   rel. difference of the two forms' outputs: %.2e, status %d" % (float(d), int(status.item())))
