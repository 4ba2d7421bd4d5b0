"""Debug aid: block statistics of the pair kernel vs the two-launch path, error pattern by block row and column."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import hiplib
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cin, cmid, cout = 512, 512, 1536
g = torch.Generator(device="cpu"); g.manual_seed(0)
w1 = (torch.randn((1, cin, cmid), generator=g) / cin ** 0.5).to(dev); w2 = (torch.randn((1, cmid, cout), generator=g) / cmid ** 0.5).to(dev)
x = torch.randn((R, cin), generator=g).to(dev)
xs = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, xs)
b1 = torch.zeros(cmid, device=dev); b2 = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
act = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hs = hiplib.SplitBuf(R, cmid, dev)
blkA = torch.full((hiplib.block_stats_floats(R, cout),), float("nan"), device=dev); blkB = torch.full_like(blkA, float("nan"))
hiplib.tdnn_layer3(xs, R, hiplib.pack_weights_bf16x3(w1), b1, None, None, act, None, 1, rv, hs)
hiplib.tdnn_layer_pool(hs, R, hiplib.pack_weights_bf16x3(w2), b2, None, None, act, None, 1, rv, blkB)
pair = hiplib.pack_pair_bf16x3(w1[0], w2[0])
hiplib.tdnn_pair_pool(xs, R, pair, (b1, None, None, None), (b2, None, None, None), act, rv, blkA)
torch.cuda.synchronize()
A = blkA.cpu().numpy().reshape(-1, 2, cout); B = blkB.cpu().numpy().reshape(-1, 2, cout)
# fp64 reference of the block means
h = torch.relu(x.double().cpu() @ w1[0].double().cpu()) if act == 1 else x.double().cpu() @ w1[0].double().cpu()
y = h @ w2[0].double().cpu()
if act == 1: y = torch.relu(y)
ref = y.numpy()[: (R // 8) * 8].reshape(-1, 8, cout).mean(axis=1)
nb = ref.shape[0]
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
print("two-launch mean vs fp64:", rel(B[:nb, 0], ref), " pair mean vs fp64:", rel(A[:nb, 0], ref))
err = np.abs(A[:nb, 0] - ref)
print("error by block (first 16):", np.round(err.mean(axis=1)[:16], 5))
print("error by column group of 64 (first 24):", np.round(err.reshape(nb, -1, 64).mean(axis=(0, 2)), 5))
print("error by column within 16:", np.round(err.reshape(nb, -1, 16).mean(axis=(0, 1)), 5))
print("M2 rel:", rel(A[:nb, 1], B[:nb, 1]))
# which 32-channel block of the first / second contraction is wrong?
def run(xm, w1m, w2m):
    xs2 = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(xm, xs2)
    pr = hiplib.pack_pair_bf16x3(w1m, w2m)
    bl = torch.full_like(blkA, float("nan"))
    hiplib.tdnn_pair_pool(xs2, R, pr, (b1, None, None, None), (b2, None, None, None), 0, rv, bl)
    torch.cuda.synchronize()
    got = bl.cpu().numpy().reshape(-1, 2, cout)[:nb, 0]
    want = ((xm.double().cpu() @ w1m.double().cpu()) @ w2m.double().cpu()).numpy()[: nb * 8].reshape(-1, 8, cout).mean(axis=1)
    return rel(got, want)
e1, e2 = [], []
for kb in range(16):
    xm = torch.zeros_like(x); xm[:, 32 * kb:32 * kb + 32] = x[:, 32 * kb:32 * kb + 32]
    e1.append(run(xm, w1[0], w2[0]))
    w2m = torch.zeros_like(w2[0]); w2m[32 * kb:32 * kb + 32] = w2[0][32 * kb:32 * kb + 32]
    e2.append(run(x, w1[0], w2m))
print("phase-1 k-block errors:", np.round(e1, 4))
print("phase-2 k-block errors:", np.round(e2, 4))
