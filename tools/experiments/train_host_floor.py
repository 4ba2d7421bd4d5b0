"""How much of the training step is the HOST: bench.py's bf16x3 step with every kernel-launching entry point of libxvector_hip.so replaced by a
Python no-op (ctypes argument marshalling, torch allocations, events, stream waits and copies stay) -- the time the host needs to ENQUEUE a step.
If that is close to the step's wall clock, the step is host-bound and kernel work removed from it buys nothing.   python tools/experiments/train_host_floor.py"""
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


print("the step as it is:            %.4f ms" % run(), flush=True)
lib = hiplib.load()
keep = ("_bytes", "_floats", "_supported", "xv_version", "xv_last_error", "xv_set_tuning", "xv_supports", "_rows")
real = {}
for sym in hiplib.SYMBOLS:
    if sym.endswith(keep) or sym in keep or "supported" in sym or "supports" in sym:
        continue
    real[sym] = getattr(lib, sym)
    setattr(lib, sym, (lambda *a: 0))
try:
    print("every launch a host no-op:    %.4f ms   (%d entry points stubbed)" % (run(), len(real)), flush=True)
    print("again:                        %.4f ms" % run(), flush=True)
finally:
    for sym, fn in real.items():
        setattr(lib, sym, fn)
print("the step as it is, again:     %.4f ms" % run(), flush=True)
