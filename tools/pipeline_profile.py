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
_blocks = kaldi_io.scan_mat_ark_windows
def blocks(fd, take_arena, first_fill=None):
    it = _blocks(fd, take_arena, first_fill)
    while True:
        t0 = time.perf_counter()
        try:
            item = next(it)
        except StopIteration:
            return
        finally:
            T["ark read + scan [reader]"] = T.get("ark read + scan [reader]", 0.0) + time.perf_counter() - t0
        yield item
kaldi_io.scan_mat_ark_windows = blocks
# GPU-side span: first frame_level launch .. last chunk_average, and wall-clock marks of the main thread
_fl = engine.DeviceModel.frame_level
marks = {}
def fl(self, *a, **k):
    if "gpu_first" not in marks:
        marks["gpu_first"] = torch.cuda.Event(enable_timing=True); marks["gpu_first"].record(); marks["t_first_launch"] = time.perf_counter()
    return _fl(self, *a, **k)
engine.DeviceModel.frame_level = fl
_fin = engine.Extractor.finish
def fin(self, h, as_array=False):
    r = _fin(self, h, as_array)
    marks["gpu_last"] = torch.cuda.Event(enable_timing=True); marks["gpu_last"].record(); marks["t_last_finish"] = time.perf_counter()
    return r
engine.Extractor.finish = fin
def stamp(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        marks.setdefault(label + " start", time.perf_counter())
        try:
            return f(*a, **k)
        finally:
            marks.setdefault(label + " end", time.perf_counter())
    setattr(obj, name, g)
stamp(models.Model, "load_model", "load_model")
stamp(engine.Extractor, "submit", "first submit")
stamp(engine.Extractor, "_staging", "first _staging")
stamp(engine.DeviceModel, "reserve", "first reserve")
stamp(kaldi_io, "arena_acquire", "first arena_acquire")
log = logging.getLogger("p"); log.addHandler(logging.NullHandler())
if os.environ.get("XV_FIRST_WINDOW"):
    models.Model.first_window_frames = int(os.environ["XV_FIRST_WINDOW"])
m = models.Model()
m.make_embedding(io.BytesIO(raw[:cut]), io.BytesIO(), mdir, 25, 10000, True, log)       # warm-up
T.clear(); marks.clear()
t0 = time.perf_counter()
out = io.BytesIO()
m.make_embedding(io.BytesIO(raw), out, mdir, 25, 10000, True, log)
tot = time.perf_counter() - t0
print("n=%d total %.3f s (%.0f utt/s)" % (n, tot, n / tot))
torch.cuda.synchronize()
print("  marks (s after start): " + ", ".join("%s %.3f" % (k, v - t0) for k, v in sorted(marks.items(), key=lambda kv: kv[1] if isinstance(kv[1], float) else 1e9) if isinstance(v, float)))
print("  first launch at %.3f s, last finish returned at %.3f s, GPU span first launch -> after last finish %.3f s" %
      (marks["t_first_launch"] - t0, marks["t_last_finish"] - t0, marks["gpu_first"].elapsed_time(marks["gpu_last"]) * 1e-3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-36s %.3f s" % (k, v))
