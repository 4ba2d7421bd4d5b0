"""Margin of the accuracy guard (engine.select_model / Extractor accuracy probe): for a range of seeds, trained-like and hostile
checkpoints of a model class -- the load-time probe value, the run-time probe value on MFCC-like utterances, and the error of every
arithmetic against the fp64 oracle when FORCED.  Prints one line per checkpoint.   python tools/guard_sweep.py [class] [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
from oracle import oracle                                      # noqa: E402  (the checker)
from xvector_amd import engine, synthetic, topology           # noqa: E402


def main():
    cls = sys.argv[1] if len(sys.argv) > 1 else "ModelWithoutDropout"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    topo = topology.get(cls)
    oracle.build()
    print("%-13s %4s | %9s %9s | %9s %8s | %9s %9s %9s | selected" % ("weights", "seed", "load", "b3|fp32", "run-time", "demoted",
                                                                       "f16bf8", "bf16x3", "fp32"))
    # "trained": checkpoints that were TRAINED by the product's own training step (synthetic.trained_checkpoint) -- seeds, step counts
    # (100 ... 1200) and learning rates vary, so that the table holds barely trained and long-trained models alike
    kinds = ("trained", "trained_like", "hostile") if cls == "ModelWithoutDropout" else ("trained_like", "hostile")
    for kind in kinds:
        for seed in range(100, 100 + n):
            if kind == "trained":
                steps = (100, 300, 600, 1200)[seed % 4]
                w, info = synthetic.trained_checkpoint(topo, 23, n_spk=64, steps=steps, learning_rate=(1e-3, 3e-3)[(seed >> 2) & 1], seed=seed)
                print("#   trained %d: %d Adam steps, loss %.3f -> %.4f, accuracy %.2f" % (seed, steps, info["first_loss"], info["last_loss"],
                                                                                          info["accuracy_last"]), flush=True)
            else:
                w = getattr(synthetic, kind)(topo, 23, seed=seed)
            mats = synthetic.mfcc_like([30, 64, 150, 256, 400, 777], 23, seed=seed + 1)
            refs = [oracle.embed_utterance(m, w, topo, 25, 10000, np.float64) for m in mats]

            def worst(model, probe):
                ex = engine.Extractor(model, 25, 10000, accuracy_probe=probe)
                got = ex.extract(mats)
                return max(oracle.rel_l2(g, r) for g, r in zip(got, refs)), ex
            sel = engine.select_model(w, topo, "cuda:0", precision="f16bf8").selection
            f8, _ = worst(engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8"), False)
            _, exg = worst(engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8"), True)
            b3, _ = worst(engine.DeviceModel(w, topo, "cuda:0", precision="bf16x3"), False)
            f32, _ = worst(engine.DeviceModel(w, topo, "cuda:0", precision="fp32"), False)
            print("%-13s %4d | %9.2e %9s | %9.2e %8s | %9.2e %9.2e %9.2e | %s" % (
                kind, seed, sel["f16bf8_vs_bf16x3"], "%.2e" % sel["bf16x3_vs_fp32"] if "bf16x3_vs_fp32" in sel else "-",
                exg.stats.get("probe_rel_l2_max", float("nan")), exg.demoted, f8, b3, f32, sel["selected"]), flush=True)


if __name__ == "__main__":
    main()
