"""Throughput of the front-end kernel (sliding-window CMN + VAD scatter) on BASELINE configs[1]-sized input (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import hiplib, synthetic
dev = torch.device("cuda:0")
lens = synthetic.utterance_lengths(10000, 200, 400, 1234).astype(np.int64)
starts = np.zeros(len(lens), np.int64); np.cumsum(lens[:-1], out=starts[1:])
total = int(lens.sum()); F = 23
rng = np.random.default_rng(0)
vad = rng.random(total) < 0.8
dst = np.full(total, -1, np.int32); dst[vad] = np.arange(int(vad.sum()), dtype=np.int32)
x = torch.randn((total, F), device=dev) * 3
y = torch.zeros((int(vad.sum()), 24), device=dev)
us, ul, d = (torch.from_numpy(a.astype(np.int32)).to(dev) for a in (starts, lens, dst))
for _ in range(3): hiplib.cmn_sliding_scatter(x, us, ul, len(lens), int(lens.max()), 300, True, 100, d, y)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): hiplib.cmn_sliding_scatter(x, us, ul, len(lens), int(lens.max()), 300, True, 100, d, y)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
by = total * F * 4 + total * 4 + int(vad.sum()) * F * 4
print("cmn_sliding_scatter: %d utts, %d frames in, %d out: %.3f ms -> %.1f M frames/s, %.0f GB/s algorithmic (in + dst_row + out)" %
      (len(lens), total, int(vad.sum()), ms, total / ms / 1e3, by / ms / 1e6))
