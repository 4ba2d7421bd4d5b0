"""Where the start-up of an extraction process goes (fresh interpreter): timestamps around every step between `python` and the
first kernel of the first window.   python tools/startup_profile.py"""
import os
import sys
import time

t_start = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
from xvector_amd import jobclock                                  # noqa: E402

marks = [("python -> script", t_start - (jobclock.process_birth() or t_start))]
t = [time.time()]


def lap(name):
    now = time.time()
    marks.append((name, now - t[0]))
    t[0] = now


import numpy as np                                                 # noqa: E402
lap("import numpy")
from xvector_amd import engine, hiplib, synthetic, topology       # noqa: E402
lap("import xvector_amd (no torch)")
topo = topology.get("ModelWithoutDropout")
w = synthetic.trained_like(topo, 23, seed=1)
lap("synthetic weights (not part of a job)")
import torch                                                       # noqa: E402
lap("import torch")
torch.cuda.init()
x = torch.zeros(16, device="cuda:0")
torch.cuda.synchronize()
lap("torch.cuda init + first allocation")
lib = hiplib.load()
lap("dlopen libxvector_hip.so")
hiplib.require_gpu()
a = torch.ones(8, device="cuda:0")
hiplib.fold_bn(a, a, a, a, 1e-3)
torch.cuda.synchronize()
lap("first kernel launch (code object load)")
model = engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8")
torch.cuda.synchronize()
lap("DeviceModel f16bf8 (upload + pack)")
twin = model.fallback()
torch.cuda.synchronize()
lap("bf16x3 twin (upload + pack)")
batch = engine.probe_batch(model.feat_dim, model.in_dim, model.gap, model.align)
lap("probe batch (host)")
model.probe_vectors(batch)
lap("probe forward f16bf8")
twin.probe_vectors(batch)
lap("probe forward bf16x3")
m2 = engine.select_model(w, topo, "cuda:0", precision="f16bf8")
lap("select_model again (warm)")
ex = engine.Extractor(m2, 25, 10000)
mats = [np.zeros((300, 23), np.float32)] * 64
ex.extract(mats)
lap("first window through the Extractor (staging buffers pinned)")
ex.extract(mats)
lap("second window")
for name, dt in marks:
    print("%-62s %8.3f s" % (name, dt))
print("%-62s %8.3f s" % ("total", sum(dt for _, dt in marks)))

# ---- where the first window's time goes: a fresh Extractor on a fresh model, under cProfile
if os.environ.get("STARTUP_CPROFILE") == "1":
    import cProfile
    import pstats
    engine._STAGE_CACHE.clear()
    m3 = engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8")
    ex3 = engine.Extractor(m3, 25, 10000, accuracy_probe=False)
    big = [np.zeros((300, 23), np.float32)] * 2300
    pr = cProfile.Profile()
    pr.enable()
    ex3.extract(big)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
