import os, sys, time, tempfile, cProfile, pstats, logging
ROOT = os.getcwd()
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)
import torch, models
from xvector_amd import engine, synthetic, topology as tp, weights as wio
topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
mdir = tempfile.mkdtemp(); wio.save_model_dir(mdir, w, topo, "ModelWithoutDropout", 64, 23)
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
for rep in range(4):
    m = models.ModelWithoutDropout()
    t0 = time.perf_counter()
    if rep == 3:
        pr = cProfile.Profile(); pr.enable()
    m.load_model(None, mdir, logging.getLogger("x"))
    if rep == 3:
        pr.disable()
    print("load_model %d: %.1f ms" % (rep, (time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
