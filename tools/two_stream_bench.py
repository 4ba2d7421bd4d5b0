"""Do two independent batches on two HIP streams fill each other's kernel tails?  (development aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import engine, synthetic, topology as tp
dev = torch.device("cuda:0")
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
models = [engine.DeviceModel(w, topo, dev) for _ in range(2)]
lens = synthetic.utterance_lengths(10000, 200, 400, 1234); order = np.argsort(lens, kind="stable")
m0 = models[0]; gap, align = m0.gap, m0.align; lead = (gap + align - 1) // align * align
batches = []; b0 = 0
while b0 < len(order):
    rows, b1 = lead, b0
    while b1 < len(order) and (b1 == b0 or rows + int(engine.slot_rows(lens[order[b1]], gap, align)) <= 131072):
        rows += int(engine.slot_rows(lens[order[b1]], gap, align)); b1 += 1
    lay = engine.BatchLayout(lens[order[b0:b1]], gap, align)
    rv = torch.from_numpy(lay.row_valid()).to(dev)
    x = torch.randn((lay.rows, m0.in_dim), device=dev) * 3.0 * rv[:, None].float(); x[:, 23:] = 0
    batches.append(dict(x=x, rs=torch.from_numpy(lay.row_start).to(dev), rl=torch.from_numpy(lay.row_len).to(dev), rv=rv,
                        n=lay.nchunks, ml=lay.max_len, lo=b0, hi=b1, rows=lay.rows)); b0 = b1
for m in models: m.reserve(max(b["rows"] for b in batches), max(b["n"] for b in batches), 400)
P = torch.empty((10000, m0.pooled_dim), device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(nstreams):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(5):
        for i, b in enumerate(batches):
            k = i % nstreams
            with torch.cuda.stream(streams[k]):
                models[k].frame_level(b["x"], b["rs"], b["rl"], b["rv"], b["n"], b["ml"], P[b["lo"]:b["hi"]])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5
for n in (1, 2, 1, 2):
    run(n); dt = run(n); print("%d stream(s): %.2f ms per 10k utterances -> %.0f utt/s (frame-level part only)" % (n, dt * 1e3, 10000 / dt))
