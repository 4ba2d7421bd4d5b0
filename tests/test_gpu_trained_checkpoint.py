"""A TRAINED checkpoint through the accuracy guard ("Next round" 3): every other weight set of the suite is a draw
(synthetic.trained_like / hostile); this one is what 300 Adam steps of the product's own training step made of a reference-style
initialisation on 64 synthetic speakers -- the closest thing to the ``model_final`` of run_xvector.sh:88-107 this image can produce.
It goes through ``engine.select_model`` (load-time probe) and the ``Extractor`` (run-time probe) in all three arithmetics and is
compared with the fp64 oracle on MFCC-like utterances.  Bar: 1e-4 relative L2 (north star), whatever arithmetic the guard selects."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_trained_checkpoint_through_the_guard(oracle_mod, capsys):
    import torch
    from xvector_amd import engine, hiplib, synthetic, topology
    hiplib.require_gpu()
    topo = topology.get("ModelWithoutDropout")
    w, info = synthetic.trained_checkpoint(topo, 23, n_spk=64, steps=300, seed=3)
    assert info["last_loss"] < 0.5 * info["first_loss"], info          # it did learn (64-way softmax starts at ln 64 = 4.16)
    # BatchNorm moving statistics moved away from their initial (0, 1)
    assert abs(float(np.mean(w["frame_level_info_layer-2/variance:0"])) - 1.0) > 0.05
    lens = [25, 60, 180, 300, 411, 800]
    mats = synthetic.mfcc_like(lens, 23, seed=5) + [m for m, _ in zip(
        (x[0].astype(np.float32) for x, _ in synthetic.speaker_minibatches(3, seed=77)), range(3))]
    refs = [oracle_mod.embed_utterance(m, w, topo, 25, 10000, np.float64) for m in mats]
    rows = []
    for precision in ("f16bf8", "bf16x3", "fp32"):
        model = engine.select_model(w, topo, "cuda:0", precision=precision)
        sel = model.selection
        ex = engine.Extractor(model, 25, 10000)
        vecs = ex.extract(mats)
        worst = max(oracle_mod.rel_l2(v, r) for v, r in zip(vecs, refs))
        rows.append((precision, sel["selected"], sel.get("f16bf8_vs_bf16x3"), ex.stats.get("probe_rel_l2_max"), bool(ex.demoted), worst))
        assert np.all(np.isfinite(np.asarray(vecs)))
        assert worst < 1e-4, rows[-1]
        if precision == "fp32":
            assert sel["selected"] == "fp32" and worst < 5e-6
    with capsys.disabled():
        print("\ntrained checkpoint (300 Adam steps, loss %.3f -> %.3f):" % (info["first_loss"], info["last_loss"]))
        for r in rows:
            print("  requested %-7s selected %-7s load-time probe %s run-time probe %s demoted %s  vs fp64 oracle %.3e" % (
                r[0], r[1], "%.3e" % r[2] if r[2] is not None else "-", "%.3e" % r[3] if r[3] is not None else "-", r[4], r[5]))
    # the guard's decision for the fast arithmetic is consistent with its own limit
    f = rows[0]
    if f[1] == "f16bf8":
        assert f[2] <= engine.PROBE_LIMIT_F16BF8
    else:
        assert f[2] > engine.PROBE_LIMIT_F16BF8 or f[2] != f[2]
