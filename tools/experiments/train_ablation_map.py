"""TIMING-ONLY ablation map of the training step (the numbers it computes are wrong): bench.py's bf16x3 step (configs[4] shapes, 200 steps after 20)
with ONE host-side call of the trainer turned into a no-op at a time -- the marginal cost of that call's kernels in the step as it runs (two
streams, the host a step ahead), which a kernel trace's durations do not give.   python tools/experiments/train_ablation_map.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd")]
sys.argv = sys.argv[:1] + ["--mode", "train", "--train-precision", "bf16x3", "--steps", "200", "--warmup", "20"]
import torch
import bench
from xvector_amd import hiplib, topology as tp
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def run():
    return bench._train_run(args, 0, 1, dev, tp.get("ModelWithoutDropout"), 23, "bf16x3", args.steps, args.warmup)["ms_per_step"]


base = [run() for _ in range(2)]
print("baseline %.4f %.4f ms/step" % tuple(base), flush=True)
b = sum(base) / 2
for name in ("rows_affine", "stats_pool", "bn_moments_fold", "bn_act_backward_parts", "pool_bn_act_backward", "adam", "ema", "pack_minibatch",
             "minibatch_layout", "bn_small_forward", "bn_small_backward", "l2_normalize_rows", "l2_normalize_backward", "softmax_ce", "am_margin",
             "axpy", "sumsq", "fc_splitk", "wgrad", "tdnn_layer3_sums", "tdnn_layer3_moments", "tdnn_layer", "merge_moments", "fold_bn", "chunk_moments"):
    real = getattr(hiplib, name)
    setattr(hiplib, name, lambda *a, **k: None)
    try:
        t = run()
        print("without %-24s %.4f ms/step  (%+.1f us, %+.1f %%)" % (name, t, (t - b) * 1e3, 100 * (t - b) / b), flush=True)
    except Exception as e:
        print("without %-24s failed: %s" % (name, str(e)[:120]), flush=True)
    finally:
        setattr(hiplib, name, real)
print("baseline again %.4f ms/step" % run())
