"""The pair kernel (layers 3 + 4 + pooling statistics): bf16x3, 16 frames per wave (xv_pair.hip) against f16bf8 with the
reduction split over a pair of waves (xv_pair8.hip); interleaved rounds.  argv[1] = rows (default 262144)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ZERO = os.environ.get("PAIR8_BENCH_ZERO", "")      # "w", "x" or "wx": zero weights / frames (how much of the time is the power limit)
cin, cmid, cout = 512, 512, int(os.environ.get('PAIR8_BENCH_COUT', '1536'))
w1 = torch.randn((cin, cmid), device=dev) / cin ** 0.5; w2 = torch.randn((cmid, cout), device=dev) / cmid ** 0.5
if 'w' in ZERO: w1.zero_(); w2.zero_()
p3, p8 = hiplib.pack_pair_bf16x3(w1, w2), hiplib.pack_pair_f16bf8(w1, w2)
x = torch.relu(torch.randn((R, cin), device=dev)) * 1.3 - 0.4
if 'x' in ZERO: x.zero_()
x3 = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, x3)
x8 = hiplib.SplitBuf(R, cin, dev, hiplib.FMT_SPLIT8); hiplib.split_encode(x, x8)
b1 = torch.zeros(cmid, device=dev); b2 = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
blk3 = torch.empty(hiplib.block_stats_floats(R, cout), device=dev); blk8 = torch.empty_like(blk3)
status = torch.zeros(1, dtype=torch.int32, device=dev)
fns = {"bf16x3, 16 frames/wave": lambda: hiplib.tdnn_pair_pool(x3, R, p3, (b1, None, None, None), (b2, None, None, None), 1, rv, blk3),
       "f16bf8, wave pairs": lambda: hiplib.tdnn_pair_pool8(x8, R, p8, (b1, None, None, None), (b2, None, None, None), 1, rv, blk8, status)}
times = {k: [] for k in fns}
for rnd in range(6):
    for name, fn in fns.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8): fn()
        b.record(); torch.cuda.synchronize()
        if rnd: times[name].append(a.elapsed_time(b) / 8)
for name, t in times.items():
    t = sorted(t); med = t[len(t) // 2]
    print("layers 3+4+pool %-24s median %.3f ms (min %.3f)  %.0f TF algorithmic" % (name, med, t[0], 2.0 * R * cin * (cmid + cout) / 1e9 / med))
a, b = blk3.view(-1, 2, cout), blk8.view(-1, 2, cout)
print("block statistics, f16bf8 vs bf16x3: mean rel L2 %.2e, M2 rel L2 %.2e, status %d" %
      (float((a[:, 0] - b[:, 0]).norm() / a[:, 0].norm()), float((a[:, 1] - b[:, 1]).norm() / a[:, 1].norm()), int(status.item())))
