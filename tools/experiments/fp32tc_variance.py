import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import engine, hiplib, synthetic, topology as tp
dev = torch.device("cuda:0")
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
R = 262144
for prec in ("fp32tc", "fp32", "fp32tc"):
    m = engine.DeviceModel(w, topo, dev, precision=prec)
    lay = engine.BatchLayout([300] * 850, m.gap, m.align)
    x = torch.randn((lay.rows, m.in_dim), device=dev) * 3; rv = torch.from_numpy(lay.row_valid()).to(dev); x *= rv[:, None].float(); x[:, 23:] = 0
    rs, rl = torch.from_numpy(lay.row_start).to(dev), torch.from_numpy(lay.row_len).to(dev)
    P = torch.empty((lay.nchunks, m.pooled_dim), device=dev)
    ts = []
    for it in range(14):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m.frame_level(x, rs, rl, rv, lay.nchunks, lay.max_len, P); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(prec, "rows", lay.rows, "ms per batch:", " ".join("%.2f" % t for t in ts))
