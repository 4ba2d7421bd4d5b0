"""Development aid: ark -> ark rate of Model.make_embedding for several read-arena sizes (first pass = cold arenas, i.e.
what a one-shot CLI call sees; later passes = warm)."""
import io, logging, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import torch
import kaldi_io, models
from xvector_amd import synthetic, topology as tp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
lens = synthetic.utterance_lengths(n, 200, 400, 4321)
rng = np.random.default_rng(4321)
pool = [(rng.standard_normal((400, 23)) * 3.0).astype(np.float32) for _ in range(257)]
bio = io.BytesIO()
cut = 0
for i in range(n):
    kaldi_io.write_mat(bio, pool[i % 257][:lens[i]], key="utt%07d" % i)
    if i == n // 20: cut = bio.tell()
raw = bio.getvalue()
d = tempfile.mkdtemp()
models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=64, feat_dim=23), d, None)
log = logging.getLogger("e2e"); log.setLevel(logging.ERROR)
models.Model().make_embedding(io.BytesIO(raw[:cut]), io.BytesIO(), d, 25, 10000, False, log)     # kernels loaded, staging pinned
for mb, first in ((48, 48), (32, 32), (64, 64), (96, 96), (144, 48), (48, 48), (64, 64), (40, 40)):
    models.Model.arena_bytes, models.Model.first_arena_bytes = mb << 20, first << 20
    kaldi_io._ARENA_FREE.clear()
    res = []
    for rep in range(3):
        out = io.BytesIO(); t0 = time.time()
        models.Model().make_embedding(io.BytesIO(raw), out, d, 25, 10000, False, log); res.append(n / (time.time() - t0))
    print("arena %3d MB first %3d MB: cold %.0f utt/s, warm %.0f / %.0f utt/s" % (mb, first, res[0], res[1], res[2]))
