"""Per-kernel timing of one ragged batch (development aid; bench.py is the judged harness)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import engine, hiplib, synthetic, topology as tp

def main():
    nutt = int(sys.argv[1]) if len(sys.argv) > 1 else 428
    prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    topo = tp.get("ModelWithoutDropout")
    w = synthetic.trained_like(topo, 23, seed=1)
    model = engine.DeviceModel(w, topo, "cuda:0", precision=prec)
    print("precision:", prec)
    lens = synthetic.utterance_lengths(nutt, 200, 400, 1234)
    layout = engine.BatchLayout(lens, model.gap)
    dev = model.device
    x = torch.randn((layout.rows, model.in_dim), device=dev) * 3; x[:, 23:] = 0
    rv = torch.from_numpy(layout.row_valid()).to(dev)
    x *= rv[:, None].float()
    rs = torch.from_numpy(layout.row_start).to(dev); rl = torch.from_numpy(layout.row_len).to(dev)
    out = torch.empty((layout.nchunks, 512), device=dev)
    model.reserve(layout.rows, layout.nchunks, layout.max_len)
    frames = int(lens.sum())
    for _ in range(2):
        model.forward_packed(x, rs, rl, rv, layout.nchunks, layout.max_len, out)
    torch.cuda.synchronize()
    # whole forward
    t0 = time.time(); n = 5
    for _ in range(n):
        model.forward_packed(x, rs, rl, rv, layout.nchunks, layout.max_len, out)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    fl = tp.flops_per_frame(topo, 23) * frames + tp.flops_per_utt(topo) * nutt
    print("batch: %d utts, %d frames, %d rows; forward %.3f ms -> %.1f utt/s, %.1f TFLOP/s (%.1f%% of 157.3)" %
          (nutt, frames, layout.rows, dt * 1e3, nutt / dt, fl / dt / 1e12, fl / dt / 157.3e12 * 100))
    if prec != "fp32":
        ref_model = engine.DeviceModel(w, topo, "cuda:0", precision="fp32")
        xr = x[:, :ref_model.in_dim].contiguous() if ref_model.in_dim != model.in_dim else x
        out_ref = torch.empty_like(out)
        ref_model.forward_packed(xr, rs, rl, rv, layout.nchunks, layout.max_len, out_ref)
        torch.cuda.synchronize()
        d = (out.double() - out_ref.double()).norm(dim=1) / out_ref.double().norm(dim=1)
        print("  %s vs fp32 x-vectors: rel-L2 max %.3e mean %.3e" % (prec, d.max().item(), d.mean().item()))
    # per kernel
    h = x; bufs = (model._ping, model._pong)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    prev = 23
    for i, L in enumerate(model.layers):
        y = model._view(model._last if i == 4 else bufs[i & 1], layout.rows, L["cout"])
        a, b = ev(), ev(); a.record()
        for _ in range(n):
            hiplib.tdnn_layer(h, L["wp"], L["bias"], L["scale"], L["shift"], 1, None, L["K"], L["dil"], rv, y)
        b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / n
        f = 2.0 * L["K"] * prev * L["cout"] * layout.rows
        print("  layer %d: %.3f ms  %.1f TFLOP/s (rows incl. gaps)" % (i, ms, f / ms / 1e9))
        h = y; prev = L["cout"]
    pooled = model._pooled[:layout.nchunks]
    a, b = ev(), ev(); a.record()
    for _ in range(n):
        hiplib.stats_pool(h, rs, rl, layout.nchunks, layout.max_len, 512, 1e-5, pooled, model._pool_ws)
    b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / n
    byts = frames * 6144 + nutt * 12288
    print("  pool: %.3f ms  %.2f TB/s (%.1f%% of 8 TB/s)" % (ms, byts / ms / 1e9, byts / ms / 1e9 / 8 * 100))
    E0 = model.embed[0]
    a, b = ev(), ev(); a.record()
    for _ in range(n):
        hiplib.fc(pooled, E0["wp"], E0["bias"], None, None, 0, None, None, out)
    b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / n
    print("  fc0: %.3f ms" % ms)

main()
