"""The accuracy guard under the reduced-precision arithmetics.

The reference computes in IEEE fp32 (local/tf/models.py:54-76).  f16bf8 (the default) and bf16x3 reproduce it to ~1e-5 / ~5e-6 on
weights that look like a trained TDNN's -- which is all ``synthetic.trained_like`` ever produced.  Here the arithmetics meet
``synthetic.hostile`` (Student-t weights, per-channel scales over three decades, BN variances 1e-4 ... 1e2, near-dead channels)
on ``synthetic.mfcc_like`` input (AR(1)-correlated, mean-normalised), against the fp64 oracle, and the two guards are exercised:

* ``engine.select_model``  -- load-time probe: a fixed batch through the LOADED weights in f16bf8 and bf16x3 (and, if that
  fails, bf16x3 vs the exact-fp32 kernels); a checkpoint whose arithmetics disagree steps down;
* ``Extractor(accuracy_probe=True)`` -- run-time probe on the caller's features: a slice of a window's first batch repeated
  on the bf16x3 twin; beyond the limit the extractor is demoted and the window repeated.
"""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BAR = 1e-4               # north star: relative L2 against the reference's fp32 CPU path
TIGHT = 3e-5             # what a checkpoint that KEEPS the f16bf8 arithmetic must deliver


@pytest.fixture(scope="module")
def env(oracle_mod):
    from xvector_amd import engine, hiplib, synthetic, topology
    hiplib.require_gpu()
    return dict(engine=engine, oracle=oracle_mod, synthetic=synthetic, topology=topology)


def _worst(env, model, w, topo, mats, probe=True):
    ex = env["engine"].Extractor(model, 25, 10000, accuracy_probe=probe)
    got = ex.extract(mats)
    refs = [env["oracle"].embed_utterance(m, w, topo, 25, 10000, np.float64) for m in mats]
    assert all(np.isfinite(g).all() for g in got)
    return max(env["oracle"].rel_l2(g, r) for g, r in zip(got, refs)), ex


def test_trained_like_checkpoints_keep_f16bf8_on_mfcc_like_input(env):
    """The comfortable case first: the probe admits f16bf8 for every model class it covers, the reported probe value is the
    typical ~1e-5, and correlated, mean-normalised MFCC-like input (not iid Gaussian) stays inside TIGHT."""
    for cls, seed in (("ModelWithoutDropout", 1), ("ModelWithoutDropoutTdnn", 2), ("ModelWithoutDropoutPRelu", 3),
                      ("ModelL2LossWithoutDropoutLRelu", 4)):
        topo = env["topology"].get(cls)
        w = env["synthetic"].trained_like(topo, 23, seed=seed)
        model = env["engine"].select_model(w, topo, "cuda:0", precision="f16bf8")
        sel = model.selection
        assert sel["probed"] and sel["selected"] == "f16bf8" and model.f16bf8, (cls, sel)
        assert sel["f16bf8_vs_bf16x3"] < env["engine"].PROBE_LIMIT_F16BF8 and not sel["probe_left_fp16_range"]
        mats = env["synthetic"].mfcc_like([40, 128, 200, 333, 400, 1000], 23, seed=seed)
        worst, ex = _worst(env, model, w, topo, mats)
        assert worst < TIGHT, (cls, worst)
        assert not ex.demoted and ex.stats["probe_windows"] == 1 and ex.stats["probe_rel_l2_max"] < env["engine"].DATA_PROBE_LIMIT
        print("%s: probe %.2e, data probe %.2e, worst vs fp64 oracle %.2e" % (cls, sel["f16bf8_vs_bf16x3"],
                                                                               ex.stats["probe_rel_l2_max"], worst))


@pytest.mark.parametrize("cls,seed", [("ModelWithoutDropout", 5), ("ModelWithoutDropout", 6), ("ModelWithoutDropout", 7),
                                      ("ModelWithoutDropoutTdnn", 8), ("ModelWithoutDropoutPRelu", 9),
                                      ("ModelL2LossWithoutDropoutLRelu", 10)])
def test_hostile_checkpoints_stay_inside_the_bar(env, cls, seed):
    """Hostile weights, MFCC-like input, through select_model + Extractor: whatever arithmetic the probe selects, every
    x-vector is within the 1e-4 bar of the fp64 oracle; a model that KEPT f16bf8 is within TIGHT, and a model that did not was
    demonstrably moved by the measured probe value.  The forced arithmetics are printed next to it (what the guard avoided)."""
    topo = env["topology"].get(cls)
    w = env["synthetic"].hostile(topo, 23, seed=seed)
    mats = env["synthetic"].mfcc_like([30, 64, 150, 256, 400, 777], 23, seed=seed + 100)
    model = env["engine"].select_model(w, topo, "cuda:0", precision="f16bf8")
    sel = model.selection
    assert sel["probed"]
    worst, ex = _worst(env, model, w, topo, mats)
    forced = {}
    for precision in ("f16bf8", "bf16x3", "fp32"):
        forced[precision] = _worst(env, env["engine"].DeviceModel(w, topo, "cuda:0", precision=precision), w, topo, mats, probe=False)[0]
    print("%s seed %d: selected %s (probe f16bf8|bf16x3 %.2e%s), guarded worst %.2e; forced: %s" % (
        cls, seed, sel["selected"], sel["f16bf8_vs_bf16x3"],
        ", bf16x3|fp32 %.2e" % sel["bf16x3_vs_fp32"] if "bf16x3_vs_fp32" in sel else "", worst,
        ", ".join("%s %.2e" % kv for kv in forced.items())))
    assert worst < BAR, (sel, worst)
    if sel["selected"] == "f16bf8" and not ex.demoted:
        assert worst < TIGHT, (sel, worst)
    else:
        assert sel["f16bf8_vs_bf16x3"] > env["engine"].PROBE_LIMIT_F16BF8 or sel["probe_left_fp16_range"] or ex.demoted
        assert model.arithmetic in ("bf16x3", "fp32tc") or ex.demoted
    assert forced["fp32"] < 3e-5                      # the exact-fp32 kernels are the reference's own arithmetic (measured: <= 1.3e-5)


def test_run_time_probe_demotes_the_extractor(env, monkeypatch):
    """The run-time probe, forced to fail (limit 0): the probed window is repeated on the bf16x3 twin, the extractor is demoted,
    later windows go to the twin directly (no second repeat) -- and every vector the caller sees is bf16x3's, bit for bit."""
    engine = env["engine"]
    topo = env["topology"].get("ModelWithoutDropout")
    w = env["synthetic"].trained_like(topo, 23, seed=21)
    mats = env["synthetic"].mfcc_like([200, 150, 90, 310, 25, 128], 23, seed=21)
    ref_ex = engine.Extractor(engine.DeviceModel(w, topo, "cuda:0", precision="bf16x3"), 25, 10000)
    ref = [ref_ex.extract(mats), ref_ex.extract(mats[:3])]
    ex = engine.Extractor(engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8"), 25, 10000)
    first = ex.extract(mats)
    assert not ex.demoted and ex.stats["probe_windows"] == 1 and ex.stats.get("fallback_windows", 0) == 0
    assert 0 < ex.stats["probe_rel_l2_max"] < engine.DATA_PROBE_LIMIT
    assert any(not np.array_equal(a, b) for a, b in zip(first, ref[0]))          # (it really ran in f16bf8)
    monkeypatch.setattr(engine, "DATA_PROBE_LIMIT", 0.0)
    ex = engine.Extractor(engine.DeviceModel(w, topo, "cuda:0", precision="f16bf8"), 25, 10000)
    h1 = ex.submit(mats)                               # window 1 is in flight when window 2 is submitted (the pipeline's order):
    h2 = ex.submit(mats[:3])                           # window 2 ran in f16bf8 before the verdict on window 1 -> repeated as well
    got = [ex.finish(h1), ex.finish(h2)]
    assert ex.demoted and ex.stats["demoted_at_window"] >= 1 and ex.stats["fallback_windows"] == 2
    h3 = ex.submit(mats)
    assert h3.get("owner") is ex._fallback_ex           # routed, not repeated
    got.append(ex.finish(h3))
    for g, r in zip(got, ref + [ref[0]]):
        for a, b in zip(g, r):
            assert np.array_equal(a, b)
    frames = sum(m.shape[0] for m in mats)
    assert ex.stats["fallback_windows"] == 2 and ex.stats["frames"] == 2 * frames + sum(m.shape[0] for m in mats[:3])


def test_load_time_probe_steps_down_and_reports(env, monkeypatch):
    """select_model with limits forced to zero walks the whole ladder f16bf8 -> bf16x3 -> fp32tc (the exact rung: exact fp32 products,
    Toom-Cook on the layers it covers; XVECTOR_EXACT_RUNG=fp32 keeps the direct contraction) and reports each measurement; with the
    probe disabled it takes the request as given."""
    engine = env["engine"]
    topo = env["topology"].get("ModelWithoutDropout")
    w = env["synthetic"].trained_like(topo, 23, seed=22)
    plain = engine.select_model(w, topo, "cuda:0", precision="f16bf8", probe=False)
    assert plain.f16bf8 and plain.selection == dict(requested="f16bf8", selected="f16bf8", probed=False)
    monkeypatch.setattr(engine, "PROBE_LIMIT_F16BF8", 0.0)
    m = engine.select_model(w, topo, "cuda:0", precision="f16bf8")
    assert m.precision == "bf16x3" and not m.f16bf8 and m.selection["selected"] == "bf16x3"
    assert 0 < m.selection["f16bf8_vs_bf16x3"] < 3e-5 and 0 < m.selection["bf16x3_vs_fp32"] < 2e-5
    monkeypatch.setattr(engine, "PROBE_LIMIT_BF16X3", 0.0)
    m = engine.select_model(w, topo, "cuda:0", precision="f16bf8")
    assert m.arithmetic == "fp32tc" and m.toom and m.selection["selected"] == "fp32tc" and m.selection["exact_rung"] == "fp32tc"
    monkeypatch.setenv("XVECTOR_EXACT_RUNG", "fp32")
    m = engine.select_model(w, topo, "cuda:0", precision="f16bf8")
    assert m.arithmetic == "fp32" and not m.toom and m.selection["selected"] == "fp32"
    monkeypatch.delenv("XVECTOR_EXACT_RUNG")
    for req in ("bf16x3", "fp32", "fp32tc"):           # nothing faster than the request is ever selected
        assert engine.select_model(w, topo, "cuda:0", precision=req).arithmetic == req


def test_make_embedding_reports_the_selected_arithmetic(env, tmp_path, monkeypatch):
    """Through the drop-in entry point: a hostile checkpoint in a model directory; Model.make_embedding logs the arithmetic it
    selected, and its output ark is within the bar of the fp64 oracle."""
    import logging
    import kaldi_io
    import models
    topo = env["topology"].get("ModelWithoutDropout")
    w = env["synthetic"].hostile(topo, 23, seed=5)
    mdir = str(tmp_path / "hostile")
    models.Model.save_model(dict(weights=w, topology=topo, model_class="ModelWithoutDropout", num_classes=64, feat_dim=23), mdir, None)
    mats = env["synthetic"].mfcc_like([60, 200, 333, 90], 23, seed=1)
    src = io.BytesIO()
    for i, m in enumerate(mats):
        kaldi_io.write_mat(src, m, key="h%d" % i)
    records = []

    class Keep(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())
    log = logging.getLogger("test_hostile_cli")
    log.setLevel(logging.INFO)
    log.addHandler(Keep())
    monkeypatch.delenv("XVECTOR_PRECISION", raising=False)
    out = io.BytesIO()
    model = models.Model()
    model.make_embedding(io.BytesIO(src.getvalue()), out, mdir, 25, 10000, False, log)
    got = dict(kaldi_io.read_vec_flt_ark(io.BytesIO(out.getvalue())))
    assert any(r.startswith("GEMM arithmetic: ") for r in records), records
    assert model.last_stats["selection"]["probed"]
    for i, m in enumerate(mats):
        assert env["oracle"].rel_l2(got["h%d" % i], env["oracle"].embed_utterance(m, w, topo, 25, 10000, np.float64)) < BAR
