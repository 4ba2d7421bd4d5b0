import cProfile, io, logging, os, pstats, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import kaldi_io, models
from xvector_amd import engine, synthetic, topology as tp
n = 10000
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
utts = synthetic.make_utterances(n, 200, 400, 23, 1234)
bio = io.BytesIO()
for k, m in utts: kaldi_io.write_mat(bio, m, key=k)
raw = bio.getvalue()
t0 = time.time(); got = [(k, np.ascontiguousarray(m, dtype=np.float32)) for k, m in kaldi_io.read_mat_ark(io.BytesIO(raw))]; print("parse: %.3f s" % (time.time() - t0))
mats = [m for _, m in got]; keys = [k for k, _ in got]
model = engine.DeviceModel(w, topo, "cuda:0"); ex = engine.Extractor(model, 25, 10000)
ex.extract(mats[:500])
t0 = time.time(); v = ex.extract(mats); print("extract: %.3f s" % (time.time() - t0))
t0 = time.time(); out = io.BytesIO(); kaldi_io.write_vec_flt_batch(out, keys, v); print("write: %.3f s" % (time.time() - t0))
pr = cProfile.Profile(); pr.enable(); v = ex.extract(mats); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
