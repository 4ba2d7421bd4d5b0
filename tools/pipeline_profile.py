"""Where the ark->ark wall time goes (development aid): reader-thread busy time, main-thread queue wait, extract, write."""
import io, logging, os, sys, tempfile, threading, time, queue
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import kaldi_io, models
from xvector_amd import engine, synthetic, topology as tp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
if len(sys.argv) > 2: sys.setswitchinterval(float(sys.argv[2]))
print('switch interval', sys.getswitchinterval())
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
utts = synthetic.make_utterances(n, 200, 400, 23, 1234)
bio = io.BytesIO()
for k, m in utts: kaldi_io.write_mat(bio, m, key=k)
raw = bio.getvalue(); del utts
model = engine.DeviceModel(w, topo, "cuda:0"); ex = engine.Extractor(model, 25, 10000)
win = models.Model.window_frames
stat = dict(parse=0.0, wait=0.0, extract=0.0, write=0.0)
q = queue.Queue(maxsize=2)
def reader():
    keys, mats, frames = [], [], 0
    t0 = time.time()
    for key, mat in kaldi_io.read_mat_ark(io.BytesIO(raw)):
        keys.append(key); mats.append(np.ascontiguousarray(mat, dtype=np.float32)); frames += mat.shape[0]
        if frames >= win:
            stat["parse"] += time.time() - t0
            q.put((keys, mats)); keys, mats, frames = [], [], 0
            t0 = time.time()
    stat["parse"] += time.time() - t0
    if keys: q.put((keys, mats))
    q.put(None)
T0 = time.time()
threading.Thread(target=reader, daemon=True).start()
out = io.BytesIO()
while True:
    t0 = time.time(); item = q.get(); stat["wait"] += time.time() - t0
    if item is None: break
    keys, mats = item
    t0 = time.time(); v = ex.extract(mats); stat["extract"] += time.time() - t0
    t0 = time.time(); kaldi_io.write_vec_flt_batch(out, keys, v); stat["write"] += time.time() - t0
tot = time.time() - T0
print("n=%d total %.3f s (%.0f utt/s): reader busy %.3f | main: wait %.3f extract %.3f write %.3f" %
      (n, tot, n / tot, stat["parse"], stat["wait"], stat["extract"], stat["write"]))
