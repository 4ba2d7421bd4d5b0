"""tests/golden/numpy_tf1.py checked ON ITS OWN (no reference, no oracle, no product code): the NumPy evaluator that lets the reference's
graph code run in the build container must itself be right, or the fixtures it produces pin nothing.

* every op against a hand-written NumPy expression of its documented TF-1 definition on small arrays (SAME padding and dilation by
  explicit index arithmetic, population variance, batch-norm formula, l2_loss, softmax cross-entropy, dropout scaling)
* reverse mode against central finite differences through a graph that uses every differentiable op
* Adam against the update written out from the optimizer's documentation, two steps
* naming: variable_scope / name_scope / default_name / xw_plus_b(name=) produce the TF names the reference later looks up
* tf.cond runs only the taken branch's assigns; Saver -> import_meta_graph -> restore goes through the files
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import numpy_tf1 as tf  # noqa: E402


@pytest.fixture(autouse=True)
def fresh_graph():
    tf.reset_default_graph()
    tf.FETCH_FLOAT64[0] = True
    yield
    tf.FETCH_FLOAT64[0] = False
    tf.RUN_HOOK[0] = None
    tf.DROPOUT_LOG[:] = []


def _run(t, feed=None):
    with tf.Session() as s:
        return s.run(t, feed)


def test_conv1d_same_is_cross_correlation_with_floor_left_padding():
    rng = np.random.default_rng(0)
    for K, d in ((5, 1), (7, 1), (1, 1), (3, 2), (3, 3), (4, 1)):            # K = 4: SAME puts the odd pad on the right
        x = rng.standard_normal((2, 11, 3))
        w = rng.standard_normal((K, 3, 4))
        xp = tf.placeholder(tf.float32, [None, None, 3])
        y = tf.convolution(xp, tf.constant(w), padding="SAME", dilation_rate=[d]) if d > 1 else tf.conv1d(xp, tf.constant(w), 1, "SAME")
        got = _run(y, {xp: x})
        left = ((K - 1) * d) // 2
        want = np.zeros((2, 11, 4))
        for b in range(2):
            for t in range(11):
                for k in range(K):
                    s = t + k * d - left
                    if 0 <= s < 11:
                        want[b, t] += x[b, s] @ w[k]
        assert np.allclose(got, want, rtol=0, atol=1e-13), (K, d)


def test_moments_batchnorm_pooling_ops():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 7, 5)) * 2 + 1
    xp = tf.placeholder(tf.float32, [None, None, 5])
    m, v = tf.moments(xp, [0, 1])
    got_m, got_v = _run([m, v], {xp: x})
    flat = x.reshape(-1, 5)
    assert np.allclose(got_m, flat.mean(0)) and np.allclose(got_v, ((flat - flat.mean(0)) ** 2).mean(0))       # population variance
    m1, v1 = tf.moments(xp, 1)
    a, b = _run([m1, v1], {xp: x})
    assert a.shape == (3, 5) and np.allclose(b, x.var(axis=1))
    g, be, mu, var = (rng.standard_normal(5) for _ in range(4))
    var = np.abs(var) + 0.1
    y = tf.batch_normalization(xp, tf.constant(mu), tf.constant(var), tf.constant(be), tf.constant(g), 1e-3)
    assert np.allclose(_run(y, {xp: x}), g * (x - mu) / np.sqrt(var + 1e-3) + be, rtol=1e-13)
    assert np.allclose(_run(tf.l2_loss(xp), {xp: x}), (x ** 2).sum() / 2)
    assert np.allclose(_run(tf.leaky_relu(xp, alpha=0.2), {xp: x}), np.where(x > 0, x, 0.2 * x))
    al = rng.standard_normal(5)
    pre = tf.maximum(0.0, xp) + tf.constant(al) * tf.minimum(0.0, xp)
    assert np.allclose(_run(pre, {xp: x}), np.where(x > 0, x, al * x))
    sm = _run(tf.softmax(xp), {xp: x})
    assert np.allclose(sm.sum(-1), 1) and np.allclose(sm, np.exp(x) / np.exp(x).sum(-1, keepdims=True))
    parts = tf.split(xp, 5, axis=2)
    assert len(parts) == 5 and np.array_equal(_run(parts[3], {xp: x}), x[:, :, 3:4])


def test_softmax_cross_entropy_accuracy_and_xw_plus_b():
    rng = np.random.default_rng(2)
    logits = rng.standard_normal((4, 6)) * 3
    labels = np.eye(6)[[1, 0, 5, 5]]
    lp, yp = tf.placeholder(tf.float32, [None, 6]), tf.placeholder(tf.float32, [None, 6])
    ce = tf.softmax_cross_entropy_with_logits(logits=lp, labels=yp)
    want = -np.log(np.exp(logits)[np.arange(4), [1, 0, 5, 5]] / np.exp(logits).sum(1))
    assert np.allclose(_run(ce, {lp: logits, yp: labels}), want)
    acc = tf.reduce_mean(tf.cast(tf.equal(tf.argmax(lp, 1), tf.argmax(yp, 1)), "float"))
    assert _run(acc, {lp: labels * 5 + 0.1 * logits, yp: labels}) == 1.0
    w, b = rng.standard_normal((6, 3)), rng.standard_normal(3)
    with tf.variable_scope("embed_layer-0"):
        s = tf.xw_plus_b(lp, tf.constant(w), tf.constant(b), name="scores")
    assert s.name == "embed_layer-0/scores:0"
    assert np.allclose(_run(s, {lp: logits}), logits @ w + b)


def test_dropout_scales_kept_elements_and_logs_the_mask():
    x = np.ones((50, 40))
    xp, kp = tf.placeholder(tf.float32, [None, 40]), tf.placeholder(tf.float32)
    y = tf.dropout(xp, kp)
    out = _run(y, {xp: x, kp: 0.8})
    name, mask, keep = tf.DROPOUT_LOG[-1]
    assert keep == 0.8 and set(np.unique(out)) <= {0.0, 1.25} and np.array_equal(out, mask / 0.8)
    assert 0.7 < mask.mean() < 0.9
    assert np.array_equal(_run(y, {xp: x, kp: 1.0}), x)                       # keep_prob = 1: the identity (models.py:412)


def _toy_graph():
    """A graph through every differentiable op the reference's classes use."""
    x = tf.placeholder(tf.float32, [None, None, 3], name="input_x")
    y = tf.placeholder(tf.float32, [None, 4], name="input_y")
    phase = tf.placeholder(tf.bool_, name="phase")
    keep = tf.placeholder(tf.float32, name="keep")
    rng = np.random.default_rng(3)
    h = x
    prev = 3
    for i, (K, d, act) in enumerate(((3, 1, "relu"), (3, 2, "lrelu"), (1, 1, "prelu"))):
        with tf.variable_scope("layer-%d" % i):
            w = tf.Variable(tf.constant(rng.standard_normal((K, prev, 6)) * 0.4), name="w")
            b = tf.Variable(tf.constant(rng.standard_normal(6) * 0.1), name="b")
            c = tf.conv1d(h, w, 1, "SAME") if d == 1 else tf.convolution(h, w, padding="SAME", dilation_rate=[d])
            h = tf.bias_add(c, b)
            if act == "relu":
                h = tf.relu(h)
            elif act == "lrelu":
                h = tf.leaky_relu(h, alpha=0.2)
            else:
                with tf.variable_scope(None, default_name="prelu"):
                    al = tf.get_variable("prelu", shape=h.get_shape()[-1], initializer=tf.constant_initializer(0.1))
                h = tf.maximum(0.0, h) + al * tf.minimum(0.0, h)
            gamma = tf.get_variable("gamma", h.get_shape()[-1], initializer=tf.constant_initializer(1.0))
            beta = tf.get_variable("beta", h.get_shape()[-1], initializer=tf.constant_initializer(0.0))
            pm = tf.get_variable("mean", h.get_shape()[-1], initializer=tf.constant_initializer(0.0), trainable=False)
            pv = tf.get_variable("variance", h.get_shape()[-1], initializer=tf.constant_initializer(1.0), trainable=False)
            hh = h

            def training(hh=hh, pm=pm, pv=pv, beta=beta, gamma=gamma):
                bm, bv = tf.moments(hh, [0, 1])
                a1 = tf.assign(pm, pm * 0.95 + bm * 0.05)
                a2 = tf.assign(pv, pv * 0.95 + bv * 0.05)
                with tf.control_dependencies([a1, a2]):
                    return tf.batch_normalization(hh, bm, bv, beta, gamma, 1e-3)

            def evaluation(hh=hh, pm=pm, pv=pv, beta=beta, gamma=gamma):
                return tf.batch_normalization(hh, pm, pv, beta, gamma, 1e-3)
            h = tf.cond(phase, training, evaluation)
            h = tf.dropout(h, keep)
            prev = 6
    h1, h2 = tf.split(h, 2, axis=2)
    with tf.variable_scope("attention"):
        aw = tf.Variable(tf.constant(rng.standard_normal((3, 3)) * 0.5), name="w")
        ab = tf.Variable(tf.constant(rng.standard_normal(3) * 0.1), name="b")
        av = tf.Variable(tf.constant(rng.standard_normal(3)), name="v")
        att = tf.softmax(tf.einsum("ijk,k->ij", tf.tanh(tf.bias_add(tf.einsum("ijk,kl->ijl", h1, aw), ab)), av))
    hm = tf.einsum("ijk,ij->ik", h2, att)
    hs = tf.subtract(tf.einsum("ijk,ij->ik", tf.square(h2), att), tf.square(hm))
    mean, var = tf.moments(h, 1)
    pooled = tf.concat([hm, tf.sqrt(hs + 1e-5), mean, tf.sqrt(var + 1e-5)], 1)
    with tf.variable_scope("output"):
        ow = tf.get_variable("w", shape=[18, 4], initializer=tf.xavier_initializer())
        ob = tf.Variable(tf.constant(0.1, shape=[4]), name="b")
        scores = tf.xw_plus_b(pooled, ow, ob, name="scores")
    losses = tf.softmax_cross_entropy_with_logits(logits=scores, labels=y)
    loss = tf.reduce_mean(tf.reduce_mean(losses) + 0.01 * (0.1 * tf.l2_loss(ow) + tf.l2_loss(ob)), name="loss")
    return x, y, phase, keep, loss


def test_gradients_match_finite_differences():
    x, y, phase, keep, loss = _toy_graph()
    rng = np.random.default_rng(4)
    feed = {x: rng.standard_normal((2, 9, 3)), y: np.eye(4)[[2, 0]], phase: True, keep: 1.0}
    g = tf.get_default_graph()
    with tf.Session() as s:
        s.run(tf.global_variables_initializer())
        tv = g.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES)
        assert [v.name for v in tv][:5] == ["layer-0/w:0", "layer-0/b:0", "layer-0/gamma:0", "layer-0/beta:0", "layer-1/w:0"]
        assert "layer-2/prelu/prelu:0" in [v.name for v in tv] and "layer-0/mean:0" not in [v.name for v in tv]
        for v in tv:                                           # move off the initial constants so no gradient is trivially zero
            v.value = v.value + 0.05 * rng.standard_normal(v.value.shape)
        keepstats = {v.name: v.value.copy() for v in g.get_collection(tf.GraphKeys.GLOBAL_VARIABLES)}
        s.run(loss, feed)
        grads = tf.Session.last_run.gradients(loss, tv)
        for v, gr in zip(tv, grads):
            flat = v.value.reshape(-1)
            for idx in rng.choice(flat.size, size=min(4, flat.size), replace=False):
                old = flat[idx]
                eps = 1e-6
                vals = []
                for sgn in (1, -1):
                    for u in g.get_collection(tf.GraphKeys.GLOBAL_VARIABLES):      # the train branch's assigns moved the moving statistics
                        if u is not v:
                            u.value = keepstats[u.name].copy()
                    w2 = v.value.copy().reshape(-1)
                    w2[idx] = old + sgn * eps
                    v.value = w2.reshape(v.value.shape)
                    vals.append(float(s.run(loss, feed)))
                w2[idx] = old
                v.value = w2.reshape(v.value.shape)
                fd = (vals[0] - vals[1]) / (2 * eps)
                assert abs(fd - gr.reshape(-1)[idx]) < 1e-6 * max(1.0, abs(fd)), (v.name, idx, fd, gr.reshape(-1)[idx])


def test_cond_runs_only_the_taken_branch_and_moving_average_decay():
    x, y, phase, keep, loss = _toy_graph()
    g = tf.get_default_graph()
    rng = np.random.default_rng(5)
    xv = rng.standard_normal((2, 9, 3))
    feed = {x: xv, y: np.eye(4)[[1, 3]], phase: False, keep: 1.0}
    with tf.Session() as s:
        s.run(tf.global_variables_initializer())
        pm = g.get_tensor_by_name("layer-0/mean:0")
        s.run(loss, feed)
        assert not pm.value.any()                                             # eval phase: no assign ran
        feed[phase] = True
        s.run(loss, feed)
        w, b = g.get_tensor_by_name("layer-0/w:0").value, g.get_tensor_by_name("layer-0/b:0").value
        xp = np.pad(xv, ((0, 0), (1, 1), (0, 0)))
        z = sum(xp[:, k:k + 9] @ w[k] for k in range(3)) + b
        r = np.maximum(z, 0).reshape(-1, 6)
        assert np.allclose(pm.value, 0.05 * r.mean(0))                        # 0 * 0.95 + batch * 0.05
        assert np.allclose(g.get_tensor_by_name("layer-0/variance:0").value, 0.95 + 0.05 * r.var(0))


def test_adam_two_steps_and_saver_round_trip(tmp_path):
    x, y, phase, keep, loss = _toy_graph()
    lr = tf.placeholder(tf.float32, name="learning_rate")
    opt = tf.AdamOptimizer(learning_rate=lr).minimize(loss, name="optimizer")
    g = tf.get_default_graph()
    assert g.get_operation_by_name("optimizer") is opt
    rng = np.random.default_rng(6)
    feed = {x: rng.standard_normal((2, 9, 3)), y: np.eye(4)[[2, 0]], phase: True, keep: 1.0, lr: 0.01}
    with tf.Session() as s:
        s.run(tf.global_variables_initializer())
        w = g.get_tensor_by_name("layer-1/w:0")
        w0 = w.value.copy()
        _, l0 = s.run([opt, loss], feed)
        g0 = tf.Session.last_run.aux["gradients"]["layer-1/w:0"]
        m1, v1 = 0.1 * g0, 0.001 * g0 * g0
        want1 = w0 - 0.01 * np.sqrt(1 - 0.999) / (1 - 0.9) * m1 / (np.sqrt(v1) + 1e-8)
        assert np.allclose(w.value, want1, rtol=1e-12, atol=0)
        _, l1 = s.run([opt, loss], feed)
        g1 = tf.Session.last_run.aux["gradients"]["layer-1/w:0"]
        m2, v2 = 0.9 * m1 + 0.1 * g1, 0.999 * v1 + 0.001 * g1 * g1
        want2 = want1 - 0.01 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2) * m2 / (np.sqrt(v2) + 1e-8)
        assert np.allclose(w.value, want2, rtol=1e-12, atol=0)
        assert np.allclose(g.get_tensor_by_name("layer-1/w/Adam:0").value, m2) and np.allclose(g.get_tensor_by_name("layer-1/w/Adam_1:0").value, v2)
        assert np.isclose(g.get_tensor_by_name("beta1_power:0").value, 0.9 ** 3)
        tf.Saver().save(s, str(tmp_path / "model"))
        kept = {v.name: v.value.copy() for v in g.get_collection(tf.GraphKeys.GLOBAL_VARIABLES)}
    assert sorted(os.listdir(str(tmp_path))) == ["checkpoint", "model.data-00000-of-00001", "model.index", "model.meta"]
    tf.reset_default_graph()
    with tf.Session() as s:
        saver = tf.import_meta_graph(str(tmp_path / "model.meta"))
        saver.restore(s, str(tmp_path / "model"))
        g2 = s.graph
        assert g2 is tf.get_default_graph() and g2 is not g
        for k, v in kept.items():
            assert np.array_equal(g2.get_tensor_by_name(k).value, v)
        feed2 = {g2.get_tensor_by_name(t.name): val for t, val in feed.items()}
        feed_eval = dict(feed2)
        feed_eval[g2.get_tensor_by_name("phase:0")] = False
        l2 = s.run(g2.get_tensor_by_name("loss:0"), feed_eval)
        assert np.isfinite(l2)
        assert s.run([g2.get_operation_by_name("optimizer"), g2.get_tensor_by_name("loss:0")], feed2)[0] is None


def test_fetches_are_float32_like_tensorflow_unless_asked():
    tf.FETCH_FLOAT64[0] = False
    xp = tf.placeholder(tf.float32, [None, 3])
    out = _run(xp * 2.0, {xp: np.ones((2, 3), np.float16)})
    assert out.dtype == np.float32
