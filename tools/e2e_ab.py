"""A/B of the ark -> ark pipeline settings in ONE process, interleaved (development aid): accuracy probes on/off, number of read
arenas; input as a BytesIO and as a tmpfs file.   python tools/e2e_ab.py [n_utts]"""
import io, logging, os, sys, tempfile, time, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import kaldi_io, models
from xvector_amd import synthetic, topology as tp, weights as wio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
work = tempfile.mkdtemp(dir="/dev/shm")
mdir = os.path.join(work, "nnet"); wio.save_model_dir(mdir, w, topo, "ModelWithoutDropout", 64, 23)
lens = synthetic.utterance_lengths(n, 200, 400, 4321)
rng = np.random.default_rng(4321)
pool = [(rng.standard_normal((400, 23)) * 3.0).astype(np.float32) for _ in range(257)]
fpath = os.path.join(work, "feats.ark")
with open(fpath, "wb") as f:
    for i in range(n):
        kaldi_io.write_mat(f, pool[i % 257][:lens[i]], key="utt%07d" % i)
data = open(fpath, "rb").read()
log = logging.getLogger("ab"); log.addHandler(logging.NullHandler()); log.propagate = False
configs = [("probe on, 5 arenas", "1", 5), ("probe off, 5 arenas", "0", 5), ("probe off, 4 arenas", "0", 4), ("probe on, 6 arenas", "1", 6)]
res = {}
try:
    for rep in range(4):
        for name, probe, arenas in configs:
            os.environ["XVECTOR_ACCURACY_PROBE"] = probe
            models.Model.arena_count = arenas
            for kind in ("bytes", "file"):
                out = io.BytesIO()
                t0 = time.perf_counter()
                if kind == "bytes":
                    models.Model().make_embedding(io.BytesIO(data), out, mdir, 25, 10000, True, log)
                else:
                    with open(fpath, "rb", buffering=0) as f:
                        models.Model().make_embedding(f, out, mdir, 25, 10000, True, log)
                dt = time.perf_counter() - t0
                if rep:
                    res.setdefault((name, kind), []).append(dt)
    for (name, kind), v in res.items():
        print("%-22s %-5s best %.4f s = %6.1f k utt/s   (all: %s)" % (name, kind, min(v), n / min(v) / 1e3, " ".join("%.3f" % x for x in v)))
finally:
    shutil.rmtree(work, ignore_errors=True)
