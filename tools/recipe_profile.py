"""Development aid: where the recipe's mode (feats.scp + vad.scp + sliding CMN on the device, tools/recipe_bench.py) spends its host
time -- cProfile of the calling thread and, through threading.setprofile, of the reader / writer threads of Model.make_embedding.
Usage: python tools/recipe_profile.py [n_utts]"""
import cProfile, io, logging, os, pstats, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import numpy as np, kaldi_io, models
from xvector_amd import synthetic, topology as tp, weights as wio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None); mdir = os.path.join(d, "m")
wio.save_model_dir(mdir, w, topo, "ModelWithoutDropout", 64, 23)
utts = synthetic.make_utterances(n, 200, 400, 23, 1234)
rng = np.random.default_rng(0)
with kaldi_io.TableWriter(d + "/f.ark", d + "/f.scp") as t, kaldi_io.TableWriter(d + "/v.ark", d + "/v.scp") as tv:
    for k, m in utts:
        kaldi_io.write_mat(t, m, key=k)
        kaldi_io.write_vec_flt(tv, (rng.random(m.shape[0]) < 0.8).astype(np.float32), key=k)
del utts
log = logging.getLogger("p"); log.addHandler(logging.NullHandler())


def run():
    out = io.BytesIO()
    t0 = time.perf_counter()
    models.Model().make_embedding(kaldi_io.MatScp(d + "/f.scp"), out, mdir, 25, 10000, True, log, vad_stream=kaldi_io.VecScp(d + "/v.scp"),
                                  cmn_window=300, cmn_center=True)
    return time.perf_counter() - t0


run(); print("plain run: %.3f s" % run())
profs = {}


def hook(frame, event, arg):                      # first event of a new thread: give it a profiler of its own
    name = threading.current_thread().name
    if name not in profs:
        profs[name] = cProfile.Profile()
        threading.setprofile(None); sys.setprofile(None)
        profs[name].enable()


threading.setprofile(hook)
main = cProfile.Profile(); main.enable(); dt = run(); main.disable()
threading.setprofile(None)
print("profiled run: %.3f s" % dt)
for name, pr in [("main", main)] + sorted(profs.items()):
    try:
        pr.disable()
    except Exception:
        pass
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print("==== thread", name); print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:3500])
import shutil; shutil.rmtree(d, ignore_errors=True)
