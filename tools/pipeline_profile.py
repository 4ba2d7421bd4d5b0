"""Where the ark->ark wall time goes (development aid): Model.make_embedding on an in-memory ark of n utterances with the
phases of the reader thread and of the main thread timed (monkeypatched wrappers; a few % of overhead)."""
import io, logging, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import kaldi_io, models
from xvector_amd import engine, synthetic, topology as tp, weights as wio
import tempfile
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
if len(sys.argv) > 2:
    sys.setswitchinterval(float(sys.argv[2]))
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
mdir = tempfile.mkdtemp(); wio.save_model_dir(mdir, w, topo, "ModelWithoutDropout", 64, 23)
utts = synthetic.make_utterances(n, 200, 400, 23, 1234)
bio = io.BytesIO()
cut = 0
for i, (k, m) in enumerate(utts):
    kaldi_io.write_mat(bio, m, key=k)
    if i == n // 10:
        cut = bio.tell()
raw = bio.getvalue(); del utts
T = {}
main_id = threading.get_ident()


def timed(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            key = label + (" [reader]" if threading.get_ident() != main_id else "")
            T[key] = T.get(key, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)


timed(models.Model, "load_model", "load_model")
timed(engine.BatchLayout, "pack", "pack")
timed(engine.BatchLayout, "row_valid", "row_valid")
timed(engine.DeviceModel, "frame_level", "frame_level (launches)")
timed(engine.DeviceModel, "segment_level", "segment_level")
timed(engine, "plan_chunk_table", "plan_chunk_table")
timed(engine.Extractor, "submit", "submit (total)")
timed(engine.Extractor, "finish", "finish (wait for the GPU + D2H)")
timed(engine, "matrix_addresses", "matrix_addresses")
timed(engine.Extractor, "_staging", "_staging")
class _Lib(object):
    def __init__(self, lib): self.lib = lib
    def xv_pack_rows_f32(self, *a):
        t0 = time.perf_counter(); rc = self.lib.xv_pack_rows_f32(*a); T["native pack"] = T.get("native pack", 0.0) + time.perf_counter() - t0
        return rc
engine._host_lib(); engine._HOST[0] = _Lib(engine._HOST[0])
timed(kaldi_io, "write_vec_flt_batch", "write")
import torch
_sync = torch.cuda.Event.synchronize
def sync(self):
    t0 = time.perf_counter(); _sync(self); T["event wait (staging set busy)"] = T.get("event wait (staging set busy)", 0.0) + time.perf_counter() - t0
torch.cuda.Event.synchronize = sync
_blocks = kaldi_io.read_mat_ark_blocks
def blocks(fd, alloc=None):
    it = _blocks(fd, alloc)
    while True:
        t0 = time.perf_counter()
        try:
            item = next(it)
        except StopIteration:
            return
        finally:
            T["ark scan + gather [reader]"] = T.get("ark scan + gather [reader]", 0.0) + time.perf_counter() - t0
        yield item
kaldi_io.read_mat_ark_blocks = blocks
log = logging.getLogger("p"); log.addHandler(logging.NullHandler())
if os.environ.get("XV_FIRST_WINDOW"):
    models.Model.first_window_frames = int(os.environ["XV_FIRST_WINDOW"])
m = models.Model()
m.make_embedding(io.BytesIO(raw[:cut]), io.BytesIO(), mdir, 25, 10000, True, log)       # warm-up
T.clear()
t0 = time.perf_counter()
out = io.BytesIO()
m.make_embedding(io.BytesIO(raw), out, mdir, 25, 10000, True, log)
tot = time.perf_counter() - t0
print("n=%d total %.3f s (%.0f utt/s)" % (n, tot, n / tot))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-36s %.3f s" % (k, v))
