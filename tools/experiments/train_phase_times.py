"""Where the training step's device time goes, UNPROFILED: CUDA events recorded on the main stream at the phase boundaries of the bf16x3 step
(configs[4] shapes; the host runs a step ahead, so the differences are device time): minibatch packed -> frame level forward done (stats pooling
starts) -> loss computed -> segment level backward done (pooling backward starts) -> optimizer.   python tools/experiments/train_phase_times.py"""
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
marks = []            # (name, event)


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def before(fn_name, name, first_only_per_step=False):
    real = getattr(hiplib, fn_name)

    def wrapped(*a, **k):
        mark(name)
        return real(*a, **k)
    setattr(hiplib, fn_name, wrapped)


def after(fn_name, name):
    real = getattr(hiplib, fn_name)

    def wrapped(*a, **k):
        r = real(*a, **k)
        mark(name)
        return r
    setattr(hiplib, fn_name, wrapped)


before("pack_minibatch", "0 step start (minibatch pack)")
before("stats_pool", "1 frame level forward done")
after("softmax_ce", "2 loss computed")
before("pool_bn_act_backward", "3 segment level backward done")
before("adam", "4 backward done (optimizer starts)")
after("adam", "5 optimizer done")
r = bench._train_run(args, 0, 1, dev, tp.get("ModelWithoutDropout"), 23, "bf16x3", args.steps, args.warmup)
torch.cuda.synchronize()
print("step %.4f ms (with the events in it)" % r["ms_per_step"])
# group by step: a step = from one "0" mark to the next
steps, cur = [], None
for name, e in marks:
    if name.startswith("0"):
        if cur:
            steps.append(cur)
        cur = []
    if cur is not None:
        cur.append((name, e))
steps = steps[40:]                                  # steady state
acc = {}
for a, b in zip(steps, steps[1:]):
    seq = a + [b[0]]
    for (n0, e0), (n1, e1) in zip(seq, seq[1:]):
        if n0.startswith("1") and n1.startswith("1"):
            continue                                # (stats_pool is called twice: first mark only)
        key = "%s -> %s" % (n0, n1)
        acc.setdefault(key, []).append(e0.elapsed_time(e1))
for k, v in acc.items():
    print("%-80s %8.1f us  (n = %d)" % (k, 1e3 * sum(v) / len(v), len(v)))
