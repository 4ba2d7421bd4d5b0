"""TIMING-ONLY ablation (the gradients are wrong): the training step (bf16x3, configs[4] shapes, bench.py's _train_run) with the bias-gradient column
sums (hiplib.col_sums(dz, None, db): one pass over dz per layer on the second stream) skipped -- the upper bound of what taking those sums out of the
BN-backward kernels that write dz could save (second lever).   python tools/experiments/train_bias_sums_ablation.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd")]
sys.argv = sys.argv[:1] + ["--mode", "train", "--train-precision", "bf16x3", "--steps", "300", "--warmup", "30"]
import torch
import bench
from xvector_amd import hiplib, topology as tp
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
real = hiplib.col_sums


def skipping(a, b, sum_a, sum_ab=None):
    if b is None:
        return                      # the bias sums: skipped (timing only)
    return real(a, b, sum_a, sum_ab)


for rnd in range(3):
    for name, fn in (("with the bias sums", real), ("WITHOUT them (timing only)", skipping)):
        hiplib.col_sums = fn
        r = bench._train_run(args, 0, 1, dev, tp.get("ModelWithoutDropout"), 23, "bf16x3", args.steps, args.warmup)
        print("%-28s %.4f ms/step" % (name, r["ms_per_step"]), flush=True)
hiplib.col_sums = real
