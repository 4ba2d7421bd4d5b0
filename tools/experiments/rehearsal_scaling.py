"""bench.py's 8 x 125 k job rehearsal at 1, 2, 4, 8 ranks on the one GPU of a box (125 k 25-frame utterances per rank each time): which
pieces of rank 0's breakdown grow with the number of processes that start together.   python tools/experiments/rehearsal_scaling.py [ranks ...]"""
import os, sys, tempfile, shutil, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")]
ranks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
sys.argv = sys.argv[:1]
import bench, models
from xvector_amd import synthetic, topology
args = bench.parse()
topo = topology.get("ModelWithoutDropout")
w = synthetic.trained_like(topo, 23, seed=1)
work = tempfile.mkdtemp(prefix="xv_rs_")
try:
    mdir = os.path.join(work, "nnet")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=64, feat_dim=23), mdir, None)
    for n in ranks:
        os.environ["XV_BENCH_REHEARSAL_RANKS"] = str(n)
        r = bench._job_rehearsal_leg(args, mdir, 23, None)
        b = r.get("breakdown_s_rank0", {})
        print("%d ranks: wall %.2f s | " % (n, r.get("wall_s", -1)) + ", ".join("%s %.2f" % (k.split(" (")[0], v) for k, v in b.items()) if "error" not in r else r)
        sys.stdout.flush()
finally:
    shutil.rmtree(work, ignore_errors=True)
