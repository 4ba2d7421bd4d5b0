import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import engine, hiplib, synthetic, topology as tp
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
nutt = int(sys.argv[1]) if len(sys.argv) > 1 else 430
m3 = engine.DeviceModel(w, topo, "cuda:0", precision="bf16x3"); m1 = engine.DeviceModel(w, topo, "cuda:0", precision="fp32")
lens = synthetic.utterance_lengths(nutt, 200, 400, 1234); lay = engine.BatchLayout(lens, m3.gap)
x = torch.randn((lay.rows, m3.in_dim), device="cuda:0") * 3; x[:, 23:] = 0
rv = torch.from_numpy(lay.row_valid()).cuda(); x *= rv[:, None].float()
for rep in range(3):
    o3 = m3.intermediates_packed(x, rv); o1 = m1.intermediates_packed(x, rv); torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(o3, o1)):
        bad = ~torch.isfinite(a)
        d = (a.double() - b.double())
        rel = d.norm() / b.double().norm()
        rowerr = d.abs().max(dim=1).values
        worst = torch.topk(rowerr, 5)
        print("rep %d layer %d: nan/inf %d, rel-L2 %.3e, worst rows %s err %s" % (rep, i, int(bad.sum()), rel.item(), worst.indices.tolist(), ["%.2e" % v for v in worst.values.tolist()]))
        if bad.any():
            r = torch.nonzero(bad.any(dim=1)).flatten()
            print("   bad rows: n=%d first %s ... mod128 %s" % (len(r), r[:8].tolist(), (r[:8] % 128).tolist()))
