"""EXPERIMENT: the training leg of bench.py in the order the default run uses it (fp32, then bf16x3, one process) against bf16x3 alone, with
one and with two streams (ORDER=fp32,bf16x3 XVECTOR_TRAIN_STREAMS=1|2): the order does not matter -- 3.21-3.24 ms after the fp32 leg as alone, 3.48-3.50 with one
stream; a default bench line once read 4.1 ms for this leg, two repeats of the same command 3.31 / 3.47 (two / one stream): a one-off.   python tools/experiments/train_leg_order.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
sys.argv = ["bench.py"]
import bench, torch
from xvector_amd import topology as tp
args = bench.parse()
dev = torch.device("cuda:0")
order = os.environ.get("ORDER", "fp32,bf16x3,bf16x3").split(",")
for prec in order:
    r = bench._train_run(args, 0, 1, dev, tp.get("ModelWithoutDropout"), 23, prec, 30, 3)
    print("%-7s %.3f ms/step  (streams %s)" % (prec, r["ms_per_step"], os.environ.get("XVECTOR_TRAIN_STREAMS", "2")), flush=True)
