"""Per-layer timing of the bf16x3 GEMM on random data (development aid): every layer shape of the default topology,
128-row vs 256-row workgroup tiles (XV_TUNE_TILE_ROWS), interleaved rounds in ONE process (median and min reported).
XVECTOR_HIP_LIB selects the library; argv[1] = rows per batch (default 262144)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ROUNDS, REPS = 5, 8
print("lib:", hiplib.SO_PATH, "rows:", R)
for (cin, cout, K) in ((512, 512, 1), (512, 1536, 1), (512, 512, 5), (512, 512, 7)):
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    wp = hiplib.pack_weights_bf16x3(w)
    x = torch.relu(torch.randn((R, cin), device=dev)); xs = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, xs)
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    ys = hiplib.SplitBuf(R, cout, dev)
    blk = torch.empty(hiplib.block_stats_floats(R, cout), device=dev)
    if cout == 1536:
        name, fn = "pool ", (lambda: hiplib.tdnn_layer_pool(xs, R, wp, bias, None, None, 1, None, 1, rv, blk))
    else:
        name, fn = "split", (lambda: hiplib.tdnn_layer3(xs, R, wp, bias, None, None, 1, None, 1, rv, ys))
    times = {128: [], 256: []}
    for rnd in range(ROUNDS + 1):
        for rows in (128, 256):
            hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, rows)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(REPS): fn()
            b.record(); torch.cuda.synchronize()
            if rnd: times[rows].append(a.elapsed_time(b) / REPS)
    hiplib.set_tuning(hiplib.TUNE_TILE_ROWS, 0)
    for rows in (128, 256):
        t = sorted(times[rows]); med, mn = t[len(t) // 2], t[0]
        ex = 6.0 * R * cin * cout * K / 1e9
        print("cin %d cout %4d K %d out=%s tile %3d: median %.3f ms (min %.3f)  %.0f TF executed (%.1f%% of 2.5 PF)" %
              (cin, cout, K, name, rows, med, mn, ex / med, ex / med / 25))

# the two K = 1 layers + pooling epilogue: two launches (layer 3 -> split buffer -> layer 4 with POOL epilogue) vs the
# register-chained pair kernel, interleaved rounds
cin, cmid, cout = 512, 512, 1536
w1 = torch.randn((1, cin, cmid), device=dev) / cin ** 0.5; w2 = torch.randn((1, cmid, cout), device=dev) / cmid ** 0.5
wp1, wp2 = hiplib.pack_weights_bf16x3(w1), hiplib.pack_weights_bf16x3(w2)
pair = hiplib.pack_pair_bf16x3(w1[0], w2[0])
x = torch.relu(torch.randn((R, cin), device=dev)); xs = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, xs)
b1 = torch.zeros(cmid, device=dev); b2 = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
hs = hiplib.SplitBuf(R, cmid, dev); blk = torch.empty(hiplib.block_stats_floats(R, cout), device=dev)
def two():
    hiplib.tdnn_layer3(xs, R, wp1, b1, None, None, 1, None, 1, rv, hs)
    hiplib.tdnn_layer_pool(hs, R, wp2, b2, None, None, 1, None, 1, rv, blk)
def one():
    hiplib.tdnn_pair_pool(xs, R, pair, (b1, None, None, None), (b2, None, None, None), 1, rv, blk)
times = {"two launches": [], "pair kernel": []}
for rnd in range(ROUNDS + 1):
    for name, fn in (("two launches", two), ("pair kernel", one)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS): fn()
        b.record(); torch.cuda.synchronize()
        if rnd: times[name].append(a.elapsed_time(b) / REPS)
for name, t in times.items():
    t = sorted(t); med = t[len(t) // 2]
    ex = 6.0 * R * cin * (cmid + cout) / 1e9
    print("layers 3+4+pool %s: median %.3f ms (min %.3f)  %.0f TF executed (%.1f%% of 2.5 PF)" % (name, med, t[0], ex / med, ex / med / 25))

# layer 0 (23 MFCC dims in 24 columns -> 512): the general kernel vs the kernel built for it
feat, cout0 = 24, 512
w0 = torch.randn((5, feat, cout0), device=dev) / (5 * 23) ** 0.5; w0[:, 23] = 0
x0 = torch.randn((R, feat), device=dev) * 3; x0[:, 23] = 0
wp0, first = hiplib.pack_weights_bf16x3(w0), hiplib.pack_first_bf16x3(w0)
b0 = torch.zeros(cout0, device=dev); y0 = hiplib.SplitBuf(R, cout0, dev)
variants = {"general kernel": lambda: hiplib.tdnn_layer3(x0, R, wp0, b0, None, None, 1, None, 1, rv, y0),
            "first-layer kernel": lambda: hiplib.tdnn_first(x0, R, first, b0, None, None, 1, None, 1, rv, y0)}
times = {k: [] for k in variants}
for rnd in range(ROUNDS + 1):
    for name, fn in variants.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS): fn()
        b.record(); torch.cuda.synchronize()
        if rnd: times[name].append(a.elapsed_time(b) / REPS)
for name, t in times.items():
    t = sorted(t); med = t[len(t) // 2]
    print("layer 0 %s: median %.3f ms (min %.3f)  output stream %.0f GB/s (%.1f%% of 8 TB/s)" %
          (name, med, t[0], R * cout0 * 4 / med / 1e6, R * cout0 * 4 / med / 1e6 / 80))
