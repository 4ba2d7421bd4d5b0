"""Development aid: the experimental 32-frames-per-wave pair kernel (csrc/xv_pair32.hip, entry points xv_x_*) against the
16-frame one (the ABI's xv_tdnn_pair_pool_bf16x3) -- block statistics agreement and interleaved timing."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
lib = hiplib.require_gpu()
vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.xv_x_packed_pair32_bytes.restype = ctypes.c_size_t; lib.xv_x_packed_pair32_bytes.argtypes = [ci, ci]
lib.xv_x_pack_pair32.restype = ci; lib.xv_x_pack_pair32.argtypes = [vp, vp, ci, ci, vp, vp]
lib.xv_x_tdnn_pair_pool32.restype = ci; lib.xv_x_tdnn_pair_pool32.argtypes = [vp, i64, ci, ci, vp] + [vp] * 8 + [ci, vp, vp, vp]
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
cin, cmid, cout = 512, 512, 1536
g = torch.Generator(device="cpu"); g.manual_seed(0)
w1 = (torch.randn((cin, cmid), generator=g) / cin ** 0.5).to(dev); w2 = (torch.randn((cmid, cout), generator=g) / cmid ** 0.5).to(dev)
x = torch.relu(torch.randn((R, cin), generator=g)).to(dev)
xs = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, xs)
b1 = (0.1 * torch.randn(cmid, generator=g)).to(dev); b2 = (0.1 * torch.randn(cout, generator=g)).to(dev)
s1 = (1 + 0.1 * torch.randn(cmid, generator=g)).to(dev); o1 = (0.1 * torch.randn(cmid, generator=g)).to(dev)
rvh = torch.ones(R, dtype=torch.uint8); rvh[5::97] = 0; rv = rvh.to(dev)
pair16 = hiplib.pack_pair_bf16x3(w1, w2)
wt32 = torch.empty(lib.xv_x_packed_pair32_bytes(cin, cout), dtype=torch.uint8, device=dev)
assert lib.xv_x_pack_pair32(P(w1), P(w2), cin, cout, P(wt32), None) == 0
blkA = torch.full((hiplib.block_stats_floats(R, cout),), float("nan"), device=dev); blkB = torch.full_like(blkA, float("nan"))
def k16(): hiplib.tdnn_pair_pool(xs, R, pair16, (b1, s1, o1, None), (b2, None, None, None), 1, rv, blkA)
def k32():
    rc = lib.xv_x_tdnn_pair_pool32(ctypes.c_void_p(xs.ptr), R, cin, cout, P(wt32), P(b1), P(s1), P(o1), None, P(b2), None, None, None, 1, P(rv), P(blkB), None)
    assert rc == 0, lib.xv_last_error()
k16(); k32(); torch.cuda.synchronize()
A = blkA.cpu().numpy().reshape(-1, 2, cout); B = blkB.cpu().numpy().reshape(-1, 2, cout)
nb = R // 8
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
print("blocks: mean rel %.3e, M2 rel %.3e, finite %s" % (rel(B[:nb, 0], A[:nb, 0]), rel(B[:nb, 1], A[:nb, 1]), np.isfinite(B[:nb]).all()))
times = {"16 frames/wave": [], "32 frames/wave": []}
for rnd in range(6):
    for name, fn in (("16 frames/wave", k16), ("32 frames/wave", k32)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8): fn()
        b.record(); torch.cuda.synchronize()
        if rnd: times[name].append(a.elapsed_time(b) / 8)
for name, t in times.items():
    t = sorted(t); med = t[len(t) // 2]; ex = 6.0 * R * cin * (cmid + cout) / 1e9
    print("pair kernel, %s: median %.3f ms (min %.3f)  %.0f TF executed (%.1f%% of 2.5 PF)" % (name, med, t[0], ex / med, ex / med / 25))
