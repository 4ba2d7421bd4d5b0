"""Development aid: extraction rate of every model class (SURVEY §8f-3 + the attention class) on BASELINE configs[1]-shaped
input resident in HBM (timing only; parity of every class is asserted by tests/test_gpu_forward.py).  bench.py stays the
headline (default class)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np
import torch
from xvector_amd import engine, hiplib, synthetic, topology as tp

ap = argparse.ArgumentParser()
ap.add_argument("--utts", type=int, default=10000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch-rows", type=int, default=262144)
ap.add_argument("--precision", default="f16bf8")
ap.add_argument("--classes", nargs="*", default=["ModelWithoutDropout", "ModelWithoutDropoutTdnn", "ModelWithoutDropoutPRelu",
                                                  "ModelL2LossWithoutDropoutLRelu", "ModelL2LossWithoutDropoutLReluAttention"])
args = ap.parse_args()
dev = torch.device("cuda:0")
feat = 23
for cls in args.classes:
    topo = tp.get(cls)
    weights = synthetic.trained_like(topo, feat, seed=1)
    model = engine.DeviceModel(weights, topo, dev, precision=args.precision)
    gap, align = model.gap, model.align
    lens = synthetic.utterance_lengths(args.utts, 200, 400, 1234)
    order = np.argsort(lens, kind="stable")
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    batches, b0 = [], 0
    lead = (gap + align - 1) // align * align
    while b0 < len(order):
        rows, b1 = lead, b0
        while b1 < len(order) and (b1 == b0 or rows + int(engine.slot_rows(lens[order[b1]], gap, align)) <= args.batch_rows):
            rows += int(engine.slot_rows(lens[order[b1]], gap, align)); b1 += 1
        lay = engine.BatchLayout(lens[order[b0:b1]], gap, align)
        rv = torch.from_numpy(lay.row_valid()).to(dev)
        x = torch.randn((lay.rows, model.in_dim), generator=gen, device=dev) * 3.0
        x *= rv[:, None].float(); x[:, feat:] = 0
        batches.append(dict(x=x, rs=torch.from_numpy(lay.row_start).to(dev), rl=torch.from_numpy(lay.row_len).to(dev), rv=rv,
                            n=lay.nchunks, max_len=lay.max_len, lo=b0, hi=b1, rows=lay.rows, lay=lay))
        b0 = b1
    n = len(order); frames = int(lens.sum())
    model.reserve(max(b["rows"] for b in batches), max(b["n"] for b in batches), max(b["max_len"] for b in batches))
    E = torch.empty((n, model.embed_dim), device=dev); P = torch.empty((n, model.pooled_dim), device=dev)
    seg = torch.arange(n + 1, dtype=torch.int32, device=dev); clen = torch.from_numpy(lens[order].astype(np.int32)).to(dev)
    xvec = torch.empty_like(E)

    def step():
        for b in batches:
            model.frame_level(b["x"], b["rs"], b["rl"], b["rv"], b["n"], b["max_len"], P[b["lo"]:b["hi"]])
        model.segment_level(P, E)
        hiplib.chunk_average(E, seg, clen, n, xvec)

    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert bool(torch.isfinite(xvec).all())
    fl = tp.flops_per_frame(topo, feat) * frames + tp.flops_per_utt(topo) * n
    print(json.dumps({"class": cls, "precision": args.precision, "utt_per_s": n / dt, "ms_per_step": dt * 1e3,
                      "algorithmic_tflops": fl / dt / 1e12, "gflop_per_utt": fl / n / 1e9}))
    del model, batches, E, P, xvec
    torch.cuda.empty_cache()
