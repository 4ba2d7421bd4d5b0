"""Development aid: file -> ark rate of Model.make_embedding in the recipe's mode (raw feats.scp + vad.scp + sliding CMN on the
device) with block tables (kaldi_io.MatScp / VecScp) and with the per-entry scp generators; checks that all runs write the same
bytes.  Then the same on a Kaldi-shaped directory: COMPRESSED feature matrices, a feats.scp that lists 90 % of them, the vad.scp all.
Usage: python tools/recipe_bench.py [n_utts]"""
import io, logging, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import numpy as np, kaldi_io, models
from xvector_amd import synthetic, topology as tp, weights as wio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
d = tempfile.mkdtemp(); mdir = os.path.join(d, "m"); wio.save_model_dir(mdir, w, topo, "ModelWithoutDropout", 64, 23)
utts = synthetic.make_utterances(n, 200, 400, 23, 1234)
rng = np.random.default_rng(0)
with kaldi_io.TableWriter(d + "/f.ark", d + "/f.scp") as t, kaldi_io.TableWriter(d + "/v.ark", d + "/v.scp") as tv:
    for k, m in utts:
        kaldi_io.write_mat(t, m, key=k)
        kaldi_io.write_vec_flt(tv, (rng.random(m.shape[0]) < 0.8).astype(np.float32), key=k)
del utts
log = logging.getLogger("p"); log.addHandler(logging.NullHandler())
outs = {}
for name, mf, mv in (("warm", lambda: kaldi_io.MatScp(d + "/f.scp"), lambda: kaldi_io.VecScp(d + "/v.scp")),
                     ("block tables (MatScp + VecScp)", lambda: kaldi_io.MatScp(d + "/f.scp"), lambda: kaldi_io.VecScp(d + "/v.scp")),
                     ("per-entry generators", lambda: kaldi_io.read_mat_scp(d + "/f.scp"), lambda: kaldi_io.read_vec_flt_scp(d + "/v.scp")),
                     ("block tables again", lambda: kaldi_io.MatScp(d + "/f.scp"), lambda: kaldi_io.VecScp(d + "/v.scp"))):
    out = io.BytesIO()
    t0 = time.perf_counter()
    models.Model().make_embedding(mf(), out, mdir, 25, 10000, True, log, vad_stream=mv(), cmn_window=300, cmn_center=True)
    dt = time.perf_counter() - t0
    outs[name] = out.getvalue()
    print("%-32s %d utts in %.3f s -> %.0f utt/s" % (name, n, dt, n / dt))
assert len(set(outs.values())) == 1, "outputs differ"
print("all outputs identical:", len(outs["warm"]), "bytes")

# ---- the shape of a real Kaldi data directory: compressed features, a subset feats.scp, vad.scp over everything ----
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fixture_inputs import encode_cm_record
pool = [encode_cm_record("k", synthetic.mfcc_like([int(rng.integers(200, 401))], 23, seed=i)[0])[2:] for i in range(200)]
plen = [m.shape[0] for _, m in kaldi_io.read_mat_ark(io.BytesIO(b"".join(("p%03d" % i).encode() + b" " + pool[i] for i in range(200))))]
lines = []
with open(d + "/c.ark", "wb") as f, kaldi_io.TableWriter(d + "/cv.ark", d + "/cv.scp") as tv:
    for i in range(n):
        key = "utt%07d" % i
        lines.append("%s %s:%d" % (key, d + "/c.ark", f.tell() + len(key) + 1))
        f.write(key.encode() + b" " + pool[i % 200])
        kaldi_io.write_vec_flt(tv, (rng.random(plen[i % 200]) < 0.8).astype(np.float32), key=key)
keep = [l for l in lines if rng.random() < 0.9]
open(d + "/c.scp", "wt").write("\n".join(keep) + "\n")
for rep in range(3):
    out = io.BytesIO()
    t0 = time.perf_counter()
    models.Model().make_embedding(kaldi_io.MatScp(d + "/c.scp"), out, mdir, 25, 10000, True, log, vad_stream=kaldi_io.VecScp(d + "/cv.scp"),
                                  cmn_window=300, cmn_center=True)
    dt = time.perf_counter() - t0
    print("%-32s %d utts in %.3f s -> %.0f utt/s" % ("compressed, 90 % subset table", len(keep), dt, len(keep) / dt))
