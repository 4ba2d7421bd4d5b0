"""PCIe- and I/O-inclusive throughput (development aid): host feature matrices -> host x-vectors through
Extractor.extract, and ark bytes -> ark bytes through Model.make_embedding."""
import io, logging, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import torch
import kaldi_io, models
from xvector_amd import engine, synthetic, topology as tp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
utts = synthetic.make_utterances(n, 200, 400, 23, 1234)
mats = [m for _, m in utts]
model = engine.DeviceModel(w, topo, "cuda:0")
ex = engine.Extractor(model, 25, 10000)
ex.extract(mats[:500]); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.time(); v = ex.extract(mats); dt = time.time() - t0
    print("Extractor.extract host->host: %d utts in %.3f s -> %.0f utt/s" % (n, dt, n / dt))
d = tempfile.mkdtemp()
models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=64, feat_dim=23), d, None)
bio = io.BytesIO()
for k, m in utts: kaldi_io.write_mat(bio, m, key=k)
raw = bio.getvalue()
log = logging.getLogger("e2e"); log.setLevel(logging.ERROR)
for rep in range(2):
    out = io.BytesIO(); t0 = time.time()
    models.Model().make_embedding(io.BytesIO(raw), out, d, 25, 10000, False, log); dt = time.time() - t0
    print("Model.make_embedding ark->ark (in-memory streams, incl. model load): %d utts, %.1f MB in, %.3f s -> %.0f utt/s" % (n, len(raw) / 1e6, dt, n / dt))
