"""A NumPy evaluator for the slice of the TensorFlow-1 graph API that the reference's graph code calls.

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY.  TensorFlow is not installed anywhere this project runs, so the reference's
``local/tf/models.py`` (every ``build_model``, ``load_model``, ``make_embedding``, ``train_one_iteration``, ``eval``) and
``local/tf/tf_block.py`` could never be *executed* -- only read.  ``install()`` registers this module as ``tensorflow`` in
``sys.modules``; the reference's Python then runs unmodified and decides, by itself, everything a reader could get wrong:
the op order of a layer (bias -> activation -> batch-norm), both epsilons, SAME padding and dilation arguments, the
variable names that fall out of its ``variable_scope`` / ``name_scope`` nesting, which tensor ``embed_layer-0/scores:0``
is, the train / eval branch of ``tf.cond``, the moving-average decay, the L2 terms of the loss, the dropout sites.
``tests/golden/make_golden.py`` uses it to write ``tests/golden/forward_refgraph.npz`` / ``train_refgraph.npz``.

What this module contributes is the *numerics of each op*, evaluated in float64 from the documented TensorFlow 1.x
definition cited at the op (``tf.nn.*`` / ``tf.*`` API documentation, r1.x).  A graph is recorded lazily (every API call
returns a ``Tensor`` node with a TF-style name and a static shape); ``Session.run`` evaluates the fetched nodes for a feed,
reverse-mode differentiates the loss when the ``minimize`` op is fetched, and applies ``tf.train.AdamOptimizer``'s update.
``tf.train.Saver`` writes ``model.meta`` (the pickled graph), ``model.index`` and ``model.data-00000-of-00001`` (values by
variable name) so that the reference's save -> ``import_meta_graph`` -> ``restore`` cycle really goes through the files.

Nothing in here is taken from the reference or from TensorFlow's sources; it is not shipped, not imported by the product,
and ``tests/test_numpy_tf1.py`` checks it on its own (hand-computed op examples, finite-difference gradients).
"""
import collections
import contextlib
import os
import pickle
import sys
import types

import numpy as np

F64 = np.float64
DROPOUT_LOG = []            # (tensor name, 0/1 mask, keep_prob) of every tf.nn.dropout evaluated, in order; a generator clears and reads it
RUN_HOOK = [None]           # f(fetches, run) called after every Session.run: a generator reads float64 values / gradients from run
FETCH_FLOAT64 = [False]     # Session.run returns float32 (what TF's float32 tensors fetch as) unless a generator asks for the float64 values


# ------------------------------------------------------------------------------------------------------------------------
# graph recording
# ------------------------------------------------------------------------------------------------------------------------
class TensorShape(tuple):
    def as_list(self):
        return list(self)


def _dim(d):
    """A static dimension: int or None.  The reference's attention class computes ``prev_dim /= 2`` (models.py:1038), a float under
    python 3; an integral float is accepted as the int it stands for."""
    if d is None:
        return None
    if isinstance(d, float):
        assert d == int(d), d
    return int(d)


def _shape(s):
    if s is None:
        return None
    if isinstance(s, (int, float, np.integer)):
        return TensorShape((_dim(s),))
    return TensorShape(_dim(d) for d in s)


class Tensor(object):
    __array_priority__ = 1000

    def __init__(self, op, inputs=(), attrs=None, name=None, shape=None, dtype="float32", graph=None):
        g = graph or get_default_graph()
        self.graph = g
        self.op = op
        self.inputs = list(inputs)
        self.attrs = attrs or {}
        self.op_name = g.unique_name(name if name is not None else op)
        self.static_shape = _shape(shape)
        self.dtype = dtype
        self.control = list(g.current_control())
        g.ops[self.op_name] = self

    @property
    def name(self):
        return self.op_name + ":0"

    @property
    def shape(self):
        return self.static_shape

    def get_shape(self):
        return self.static_shape

    def __repr__(self):
        return "<Tensor %s %s %s>" % (self.name, self.op, self.static_shape)

    # operator overloads (tf.Tensor overloads + - * / and unary -)
    def __add__(self, o): return add(self, o)
    def __radd__(self, o): return add(o, self)
    def __sub__(self, o): return subtract(self, o)
    def __rsub__(self, o): return subtract(o, self)
    def __mul__(self, o): return multiply(self, o)
    def __rmul__(self, o): return multiply(o, self)
    def __truediv__(self, o): return divide(self, o)
    def __rtruediv__(self, o): return divide(o, self)
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __neg__(self): return negative(self)


class Variable(Tensor):
    """tf.Variable(initial_value, name=...) / tf.get_variable(name, shape, initializer=...).  The value lives on the node and is
    dropped from the pickled graph (it travels in the checkpoint's data file, as in TF)."""

    def __init__(self, initial_value=None, trainable=True, name=None, dtype=None, _full_name=None, _initializer=None, _shape_=None):
        g = get_default_graph()
        if _full_name is None:
            init = convert(initial_value)
            shape = init.static_shape
            op_name = None
        else:
            init, shape = None, _shape(_shape_)
        Tensor.__init__(self, "variable", [], dict(trainable=bool(trainable)), name if _full_name is None else None, shape, "float32", g)
        if _full_name is not None:                      # get_variable: named by the VARIABLE scope, not the name scope
            del g.ops[self.op_name]
            assert _full_name not in g.ops, "variable %s already exists" % _full_name
            self.op_name = _full_name
            g.names_in_use[_full_name.lower()] = 1
            g.ops[_full_name] = self
        self.control = []
        self.initial_value = init
        self.initializer_fn = _initializer
        self.value = None
        g.collections[GraphKeys.GLOBAL_VARIABLES].append(self)
        if trainable:
            g.collections[GraphKeys.TRAINABLE_VARIABLES].append(self)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["value"] = None
        return d


class Operation(Tensor):
    """An op fetched for its effect (global_variables_initializer, minimize); fetching it returns None."""

    @property
    def name(self):
        return self.op_name


class GraphKeys(object):
    GLOBAL_VARIABLES = "variables"
    TRAINABLE_VARIABLES = "trainable_variables"
    UPDATE_OPS = "update_ops"


class Graph(object):
    def __init__(self):
        self.ops = collections.OrderedDict()
        self.collections = collections.defaultdict(list)
        self.names_in_use = {}
        self.name_stack = ""
        self.var_scope = ""
        self.var_scope_counts = {}
        self.control_stack = []
        self.seed = None
        self.rng_state = None

    # ---- naming: ops are named <name scope>/<name>, made unique with _1, _2 ... (tf.Graph.unique_name) ----
    def unique_name(self, name, mark=True):
        if name.endswith("/"):
            return name[:-1]
        full = self.name_stack + "/" + name if self.name_stack else name
        key = full.lower()
        n = self.names_in_use.get(key, 0)
        if mark:
            self.names_in_use[key] = n + 1
        if n > 0:
            base = full
            while True:
                full = "%s_%d" % (base, n)
                if full.lower() not in self.names_in_use:
                    break
                n += 1
            if mark:
                self.names_in_use[full.lower()] = 1
        return full

    @contextlib.contextmanager
    def name_scope(self, name):
        old = self.name_stack
        self.name_stack = self.unique_name(name) if name else ""
        try:
            yield self.name_stack + "/" if self.name_stack else ""
        finally:
            self.name_stack = old

    def current_control(self):
        out = []
        for c in self.control_stack:
            out.extend(c)
        return out

    @contextlib.contextmanager
    def as_default(self):
        _GRAPHS.append(self)
        try:
            yield self
        finally:
            _GRAPHS.pop()

    def get_tensor_by_name(self, name):
        assert name.endswith(":0"), name
        t = self.ops[name[:-2]]
        return t

    def get_operation_by_name(self, name):
        return self.ops[name]

    def get_collection(self, key):
        return list(self.collections.get(key, []))

    def adopt(self, other):
        seed = self.seed                        # tf.set_random_seed called before the import keeps its effect
        self.__dict__ = other.__dict__
        if seed is not None:
            self.seed, self.rng_state = seed, None
        for t in self.ops.values():
            t.graph = self

    def rng(self):
        if self.rng_state is None:
            self.rng_state = np.random.default_rng(self.seed if self.seed is not None else 0)
        return self.rng_state


_GRAPHS = [Graph()]


def get_default_graph():
    return _GRAPHS[-1]


def reset_default_graph():
    _GRAPHS[0] = Graph()
    if len(_GRAPHS) == 1:
        return
    raise AssertionError("reset_default_graph inside a graph context")


def set_random_seed(seed):
    g = get_default_graph()
    g.seed, g.rng_state = int(seed), None


def get_collection(key):
    return get_default_graph().get_collection(key)


def name_scope(name, default_name=None, values=None):
    return get_default_graph().name_scope(name if name is not None else default_name)


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None):
    """tf.variable_scope: prefixes tf.get_variable names with the scope and opens a name scope of the same name; with
    name_or_scope=None the scope is default_name made unique among its siblings (tf_block.py:40 -> '<layer scope>/prelu')."""
    g = get_default_graph()
    old = g.var_scope
    if name_or_scope is None:
        base = (old + "/" if old else "") + default_name
        n = g.var_scope_counts.get(base, 0)
        g.var_scope_counts[base] = n + 1
        leaf = default_name if n == 0 else "%s_%d" % (default_name, n)
    else:
        leaf = name_or_scope
    g.var_scope = (old + "/" if old else "") + leaf
    try:
        with g.name_scope(leaf):
            yield g.var_scope
    finally:
        g.var_scope = old


@contextlib.contextmanager
def control_dependencies(tensors):
    g = get_default_graph()
    g.control_stack.append(list(tensors or []))
    try:
        yield
    finally:
        g.control_stack.pop()


# ------------------------------------------------------------------------------------------------------------------------
# static shapes
# ------------------------------------------------------------------------------------------------------------------------
def _bshape(a, b):
    if a is None or b is None:
        return None
    a, b = list(a), list(b)
    n = max(len(a), len(b))
    a, b = [1] * (n - len(a)) + a, [1] * (n - len(b)) + b
    out = []
    for x, y in zip(a, b):
        if x == 1:
            out.append(y)
        elif y == 1 or x == y:
            out.append(x)
        elif x is None or y is None:
            out.append(x if y is None else y)
        else:
            raise ValueError("shapes %s and %s do not broadcast" % (a, b))
    return out


def convert(v, dtype="float32"):
    if isinstance(v, Tensor):
        return v
    a = np.asarray(v)
    return Tensor("const", [], dict(value=a.astype(F64) if a.dtype.kind in "fiu" else a), "Const", a.shape, dtype)


# ------------------------------------------------------------------------------------------------------------------------
# the API surface (constructors of graph nodes)
# ------------------------------------------------------------------------------------------------------------------------
float32, bool_ = "float32", "bool"


def placeholder(dtype, shape=None, name=None):
    return Tensor("placeholder", [], {}, name or "Placeholder", shape, dtype)


def constant(value, dtype=None, shape=None, name="Const"):
    """tf.constant(value, shape=...): a scalar value fills the shape."""
    a = np.asarray(value, F64)
    if shape is not None:
        shp = tuple(_shape(shape))
        a = np.full(shp, a, F64) if a.ndim == 0 else a.reshape(shp)
    return Tensor("const", [], dict(value=a), name, a.shape, dtype or "float32")


def zeros(shape, dtype=None, name="zeros"):
    return constant(0.0, dtype, shape, name)


def ones(shape, dtype=None, name="ones"):
    return constant(1.0, dtype, shape, name)


def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name="truncated_normal"):
    """tf.truncated_normal: normal(mean, stddev) samples; "values whose magnitude is more than 2 standard deviations from the mean are
    dropped and re-picked".  (The random stream is NumPy's, not TF's: initial values are checked by their law, never by value.)"""
    return Tensor("truncated_normal", [convert(mean), convert(stddev)], dict(shape=tuple(_shape(shape))), name, shape)


def random_uniform(shape, minval=0, maxval=None, dtype=None, seed=None, name="random_uniform"):
    """tf.random_uniform: uniform in [minval, maxval)."""
    return Tensor("random_uniform", [convert(minval), convert(1.0 if maxval is None else maxval)], dict(shape=tuple(_shape(shape))), name, shape)


class constant_initializer(object):
    """tf.constant_initializer(value): every element is value."""

    def __init__(self, value=0.0):
        self.value = float(value)

    def __call__(self, shape, rng):
        return np.full(shape, self.value, F64)


class xavier_initializer(object):
    """tf.contrib.layers.xavier_initializer(uniform=True): uniform in +-sqrt(6 / (fan_in + fan_out)) (Glorot & Bengio 2010;
    variance_scaling_initializer(factor=1, mode='FAN_AVG', uniform=True): limit = sqrt(3 * factor / n), n = (fan_in + fan_out) / 2)."""

    def __init__(self, uniform=True):
        assert uniform

    def __call__(self, shape, rng):
        fan_in = int(np.prod(shape[:-1])) if len(shape) > 1 else int(shape[0])
        lim = np.sqrt(6.0 / (fan_in + int(shape[-1])))
        return rng.uniform(-lim, lim, size=shape)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    g = get_default_graph()
    full = (g.var_scope + "/" if g.var_scope else "") + name
    return Variable(trainable=trainable, _full_name=full, _initializer=initializer, _shape_=_shape(shape))


def global_variables_initializer():
    return Operation("init", [], {}, "init", None)


def _binary(op, a, b, name):
    a, b = convert(a), convert(b)
    return Tensor(op, [a, b], {}, name, _bshape(a.static_shape, b.static_shape))


def add(a, b, name="add"): return _binary("add", a, b, name)
def subtract(a, b, name="sub"): return _binary("sub", a, b, name)
def multiply(a, b, name="mul"): return _binary("mul", a, b, name)
def divide(a, b, name="truediv"): return _binary("div", a, b, name)
def maximum(a, b, name="Maximum"): return _binary("maximum", a, b, name)
def minimum(a, b, name="Minimum"): return _binary("minimum", a, b, name)


def _unary(op, x, name, **attrs):
    x = convert(x)
    return Tensor(op, [x], attrs, name, x.static_shape)


def negative(x, name="Neg"): return _unary("neg", x, name)
def sqrt(x, name="Sqrt"): return _unary("sqrt", x, name)
def rsqrt(x, name="Rsqrt"): return _unary("rsqrt", x, name)
def square(x, name="Square"): return _unary("square", x, name)
def tanh(x, name="Tanh"): return _unary("tanh", x, name)
def stop_gradient(x, name="StopGradient"): return _unary("stop_gradient", x, name)
def identity(x, name="Identity"): return _unary("identity", x, name)


def relu(features, name="Relu"):
    """tf.nn.relu: max(features, 0)."""
    return _unary("relu", features, name)


def leaky_relu(features, alpha=0.2, name="LeakyRelu"):
    """tf.nn.leaky_relu(features, alpha): max(alpha * features, features)."""
    return _unary("leaky_relu", features, name, alpha=float(alpha))


def softmax(logits, axis=-1, name="Softmax"):
    """tf.nn.softmax: exp(logits) / reduce_sum(exp(logits), axis), last axis by default."""
    assert axis == -1
    return _unary("softmax", logits, name)


def _reduced_shape(shape, axes, keepdims):
    if shape is None:
        return None
    nd = len(shape)
    axes = list(range(nd)) if axes is None else [a % nd for a in axes]
    return [1 if i in axes else d for i, d in enumerate(shape)] if keepdims else [d for i, d in enumerate(shape) if i not in axes]


def _axes(axis):
    if axis is None:
        return None
    return [int(axis)] if isinstance(axis, (int, np.integer)) else [int(a) for a in axis]


def reduce_mean(x, axis=None, keepdims=False, name="Mean"):
    x = convert(x)
    ax = _axes(axis)
    return Tensor("reduce_mean", [x], dict(axes=ax, keepdims=keepdims), name, _reduced_shape(x.static_shape, ax, keepdims))


def reduce_sum(x, axis=None, keepdims=False, name="Sum"):
    x = convert(x)
    ax = _axes(axis)
    return Tensor("reduce_sum", [x], dict(axes=ax, keepdims=keepdims), name, _reduced_shape(x.static_shape, ax, keepdims))


def squeeze(x, axis, name="Squeeze"):
    x = convert(x)
    ax = _axes(axis)
    shp = None if x.static_shape is None else [d for i, d in enumerate(x.static_shape) if i not in [a % len(x.static_shape) for a in ax]]
    return Tensor("squeeze", [x], dict(axes=ax), name, shp)


def moments(x, axes, name=None, keep_dims=False):
    """tf.nn.moments(x, axes) -> (mean, variance): mean = reduce_mean(x, axes); variance = reduce_mean(squared_difference(x,
    stop_gradient(mean)), axes) -- the biased (population, divide-by-N) variance about the mean."""
    with name_scope(name, "moments"):
        ax = _axes(axes)
        mean = reduce_mean(x, ax, keepdims=True, name="mean")
        var = reduce_mean(square(subtract(x, stop_gradient(mean)), name="SquaredDifference"), ax, keepdims=True, name="variance")
        if keep_dims:
            return mean, var
        return squeeze(mean, ax, "Squeeze"), squeeze(var, ax, "Squeeze_1")


def batch_normalization(x, mean, variance, offset, scale, variance_epsilon, name=None):
    """tf.nn.batch_normalization: scale * (x - mean) / sqrt(variance + variance_epsilon) + offset, evaluated as documented:
    inv = rsqrt(variance + variance_epsilon) * scale;  x * inv + (offset - mean * inv)."""
    with name_scope(name, "batchnorm"):
        inv = rsqrt(add(variance, variance_epsilon))
        if scale is not None:
            inv = multiply(inv, scale)
        shift = subtract(offset, multiply(mean, inv)) if offset is not None else negative(multiply(mean, inv))
        return add(multiply(x, inv), shift, name="add_1")


def bias_add(value, bias, name="BiasAdd"):
    """tf.nn.bias_add: adds the 1-D bias along the last dimension."""
    value, bias = convert(value), convert(bias)
    return Tensor("bias_add", [value, bias], {}, name, value.static_shape)


def _conv(x, w, dilation, padding, name):
    x, w = convert(x), convert(w)
    assert padding == "SAME"
    shp = None if x.static_shape is None else list(x.static_shape[:2]) + [w.static_shape[2]]
    return Tensor("conv1d", [x, w], dict(dilation=int(dilation)), name, shp)


def conv1d(value, filters, stride, padding, name="conv1d"):
    """tf.nn.conv1d(value[batch, width, in], filters[width, in, out], stride, 'SAME'): cross-correlation (no kernel flip) of the
    NWC input; SAME with stride 1: pad_total = filter_width - 1, pad_left = pad_total // 2, the rest on the right, zeros."""
    assert stride == 1
    return _conv(value, filters, 1, padding, name)


def convolution(input, filter, padding, strides=None, dilation_rate=None, name="convolution"):
    """tf.nn.convolution(input, filter, padding='SAME', dilation_rate=[d]): atrous cross-correlation, out[b, t] = sum_k
    in[b, t + d * k - pad_left] . filter[k]; SAME pads the EFFECTIVE filter (width (K - 1) * d + 1): pad_total = (K - 1) * d,
    pad_left = pad_total // 2."""
    d = 1 if dilation_rate is None else int(dilation_rate[0])
    assert strides is None
    return _conv(input, filter, d, padding, name)


def matmul(a, b, name="MatMul"):
    a, b = convert(a), convert(b)
    return Tensor("matmul", [a, b], {}, name, [a.static_shape[0], b.static_shape[1]])


def xw_plus_b(x, weights, biases, name=None):
    """tf.nn.xw_plus_b: matmul(x, weights) + biases; the result carries the given name (the reference fetches
    '<scope>/scores:0')."""
    with name_scope(name, "xw_plus_b") as scope:
        mm = matmul(x, weights)
        return bias_add(mm, biases, name=scope)


def dropout(x, keep_prob, name=None):
    """tf.nn.dropout(x, keep_prob): "With probability keep_prob, outputs the input element scaled up by 1 / keep_prob, otherwise outputs
    0": binary = floor(keep_prob + uniform[0, 1)); x / keep_prob * binary.  The uniform stream is NumPy's; every mask drawn is
    appended to DROPOUT_LOG so a checker can be handed the same mask."""
    with name_scope(name, "dropout"):
        x = convert(x)
        return Tensor("dropout", [x, convert(keep_prob)], {}, "mul", x.static_shape)


def concat(values, axis, name="concat"):
    values = [convert(v) for v in values]
    shp = None
    if all(v.static_shape is not None for v in values):
        shp = list(values[0].static_shape)
        dims = [v.static_shape[axis] for v in values]
        shp[axis] = None if any(d is None for d in dims) else sum(dims)
    return Tensor("concat", values, dict(axis=int(axis)), name, shp)


def split(value, num_or_size_splits, axis=0, name="split"):
    """tf.split(value, n, axis): n equal slices along axis."""
    value = convert(value)
    n = int(num_or_size_splits)
    size = value.static_shape[axis] // n
    shp = list(value.static_shape)
    shp[axis] = size
    with name_scope(name):
        return [Tensor("slice", [value], dict(axis=int(axis), start=i * size, size=size), "split", shp) for i in range(n)]


def einsum(equation, *inputs, **kw):
    """tf.einsum(equation, a, b): explicit Einstein summation (indices absent from the output are summed)."""
    ins = [convert(v) for v in inputs]
    lhs, out = equation.replace(" ", "").split("->")
    subs = lhs.split(",")
    assert len(subs) == 2 and len(ins) == 2
    dims = {}
    for s, t in zip(subs, ins):
        for ch, d in zip(s, t.static_shape):
            dims[ch] = d if dims.get(ch) is None else dims[ch]
    return Tensor("einsum", ins, dict(subs=subs, out=out), kw.get("name") or "einsum", [dims[ch] for ch in out])


def l2_loss(t, name="L2Loss"):
    """tf.nn.l2_loss(t): sum(t ** 2) / 2."""
    return Tensor("l2_loss", [convert(t)], {}, name, ())


def softmax_cross_entropy_with_logits(_sentinel=None, labels=None, logits=None, dim=-1, name="softmax_cross_entropy_with_logits"):
    """tf.nn.softmax_cross_entropy_with_logits(labels, logits): per row, -sum_j labels[j] * log_softmax(logits)[j]; labels are a
    probability distribution per row; the gradient reaches the logits only (softmax - labels)."""
    assert _sentinel is None and dim == -1
    logits, labels = convert(logits), convert(labels)
    return Tensor("softmax_xent", [logits, labels], {}, name, [logits.static_shape[0]])


def argmax(input, axis=None, name="ArgMax"):
    x = convert(input)
    return Tensor("argmax", [x], dict(axis=int(axis)), name, _reduced_shape(x.static_shape, [int(axis)], False), "int64")


def equal(x, y, name="Equal"):
    x, y = convert(x), convert(y)
    return Tensor("equal", [x, y], {}, name, _bshape(x.static_shape, y.static_shape), "bool")


def cast(x, dtype, name="Cast"):
    x = convert(x)
    return Tensor("cast", [x], dict(dtype=dtype), name, x.static_shape, dtype)


def assign(ref, value, name="Assign"):
    """tf.assign(ref, value): writes value into the variable when the op runs and outputs the new value."""
    assert isinstance(ref, Variable)
    value = convert(value)
    return Tensor("assign", [value], dict(ref=ref), name, ref.static_shape)


def cond(pred, true_fn=None, false_fn=None, name="cond", fn1=None, fn2=None):
    """tf.cond(pred, true_fn, false_fn): both branches are built; only the ops of the taken branch run (so the assigns of
    tf_block.py:20-21 run in the training phase only)."""
    true_fn, false_fn = true_fn or fn1, false_fn or fn2
    with name_scope(name):
        with name_scope("true"):
            a = true_fn()
        with name_scope("false"):
            b = false_fn()
        return Tensor("cond", [convert(pred), a, b], {}, "Merge", a.static_shape)


class AdamOptimizer(object):
    """tf.train.AdamOptimizer(learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8), as its API documentation states the update:
        lr_t = learning_rate * sqrt(1 - beta2^t) / (1 - beta1^t);  m_t = beta1 * m + (1 - beta1) * g;
        v_t = beta2 * v + (1 - beta2) * g * g;  variable -= lr_t * m_t / (sqrt(v_t) + epsilon)
    with the slot variables '<var>/Adam' (m), '<var>/Adam_1' (v) and the accumulators 'beta1_power' / 'beta2_power' (beta^t, initialised
    to beta, multiplied by beta after every step) created as non-trainable global variables -- so a Saver keeps them, as TF's does."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, float(beta1), float(beta2), float(epsilon)

    def minimize(self, loss, name="Adam", var_list=None):
        g = get_default_graph()
        vs = list(var_list or g.get_collection(GraphKeys.TRAINABLE_VARIABLES))
        old_stack, g.name_stack = g.name_stack, ""
        slots = []
        try:
            for v in vs:
                m = Variable(trainable=False, _full_name=v.op_name + "/Adam", _initializer=constant_initializer(0.0), _shape_=v.static_shape)
                s = Variable(trainable=False, _full_name=v.op_name + "/Adam_1", _initializer=constant_initializer(0.0), _shape_=v.static_shape)
                slots.append((m, s))
            p1 = Variable(trainable=False, _full_name="beta1_power", _initializer=constant_initializer(self.b1), _shape_=())
            p2 = Variable(trainable=False, _full_name="beta2_power", _initializer=constant_initializer(self.b2), _shape_=())
        finally:
            g.name_stack = old_stack
        return Operation("minimize", [convert(loss), convert(self.lr)],
                         dict(vars=vs, slots=slots, powers=(p1, p2), b1=self.b1, b2=self.b2, eps=self.eps), name, None)


# ------------------------------------------------------------------------------------------------------------------------
# evaluation
# ------------------------------------------------------------------------------------------------------------------------
def _unbroadcast(g, shape):
    g = np.asarray(g)
    while g.ndim > len(shape):
        g = g.sum(axis=0)
    for i, d in enumerate(shape):
        if d == 1 and g.shape[i] != 1:
            g = g.sum(axis=i, keepdims=True)
    return g


def _conv_pad(T, K, d):
    total = (K - 1) * d
    left = total // 2
    return left, total - left


def _conv_forward(x, w, d):
    B, T, _ = x.shape
    K = w.shape[0]
    left, right = _conv_pad(T, K, d)
    xp = np.pad(x, ((0, 0), (left, right), (0, 0)))
    z = np.zeros((B, T, w.shape[2]), F64)
    for k in range(K):
        z += xp[:, k * d:k * d + T, :] @ w[k]
    return z


def _conv_backward(x, w, d, gz):
    B, T, C = x.shape
    K = w.shape[0]
    left, right = _conv_pad(T, K, d)
    xp = np.pad(x, ((0, 0), (left, right), (0, 0)))
    gxp = np.zeros_like(xp)
    gw = np.zeros_like(w)
    g2 = gz.reshape(B * T, -1)
    for k in range(K):
        gxp[:, k * d:k * d + T, :] += gz @ w[k].T
        gw[k] = xp[:, k * d:k * d + T, :].reshape(B * T, C).T @ g2
    return gxp[:, left:left + T, :], gw


class _Run(object):
    def __init__(self, graph, feed):
        self.graph = graph
        self.val = {}
        self.taken = {}
        self.aux = {}
        for k, v in (feed or {}).items():
            a = np.asarray(v)
            self.val[id(k)] = a.astype(bool) if k.dtype == "bool" else a.astype(F64)

    def get(self, t):
        key = id(t)
        if key in self.val:
            return self.val[key]
        for c in t.control:
            self.get(c)
        v = self._eval(t)
        self.val[key] = v
        return v

    def active_inputs(self, t):
        if t.op == "cond":
            return [t.inputs[1] if self.taken[id(t)] else t.inputs[2]]
        return t.inputs

    def _eval(self, t):
        op, a = t.op, t.attrs
        if op == "placeholder":
            raise KeyError("placeholder %s was not fed" % t.name)
        if op == "variable":
            assert t.value is not None, "variable %s is not initialised" % t.name
            return t.value
        if op == "const":
            return a["value"]
        if op == "cond":
            pred = bool(self.get(t.inputs[0]))
            self.taken[id(t)] = pred
            return self.get(t.inputs[1] if pred else t.inputs[2])
        if op == "init":
            for v in self.graph.get_collection(GraphKeys.GLOBAL_VARIABLES):
                if v.initial_value is not None:
                    v.value = np.array(_Run(self.graph, None).get(v.initial_value), F64)
                else:
                    v.value = np.asarray(v.initializer_fn(tuple(v.static_shape), self.graph.rng()), F64)
            return None
        if op == "minimize":
            return self._adam(t)
        x = [self.get(i) for i in t.inputs]
        if op == "add": return x[0] + x[1]
        if op == "sub": return x[0] - x[1]
        if op == "mul": return x[0] * x[1]
        if op == "div": return x[0] / x[1]
        if op == "maximum": return np.maximum(x[0], x[1])
        if op == "minimum": return np.minimum(x[0], x[1])
        if op == "neg": return -x[0]
        if op == "sqrt": return np.sqrt(x[0])
        if op == "rsqrt": return 1.0 / np.sqrt(x[0])
        if op == "square": return x[0] * x[0]
        if op == "tanh": return np.tanh(x[0])
        if op in ("stop_gradient", "identity"): return x[0]
        if op == "relu": return np.maximum(x[0], 0.0)
        if op == "leaky_relu": return np.maximum(a["alpha"] * x[0], x[0])
        if op == "softmax":
            e = np.exp(x[0] - x[0].max(axis=-1, keepdims=True))
            return e / e.sum(axis=-1, keepdims=True)
        if op == "reduce_mean": return x[0].mean(axis=None if a["axes"] is None else tuple(a["axes"]), keepdims=a["keepdims"])
        if op == "reduce_sum": return x[0].sum(axis=None if a["axes"] is None else tuple(a["axes"]), keepdims=a["keepdims"])
        if op == "squeeze": return np.squeeze(x[0], axis=tuple(a["axes"]))
        if op == "bias_add": return x[0] + x[1]
        if op == "conv1d": return _conv_forward(x[0], x[1], a["dilation"])
        if op == "matmul": return x[0] @ x[1]
        if op == "dropout":
            keep = float(x[1])
            u = self.graph.rng().random(x[0].shape)
            mask = np.floor(keep + u)
            DROPOUT_LOG.append((t.name, mask, keep))
            self.aux[id(t)] = (mask, keep)
            return x[0] / keep * mask
        if op == "concat": return np.concatenate(x, axis=a["axis"])
        if op == "slice":
            idx = [slice(None)] * x[0].ndim
            idx[a["axis"]] = slice(a["start"], a["start"] + a["size"])
            return x[0][tuple(idx)]
        if op == "einsum": return np.einsum(",".join(a["subs"]) + "->" + a["out"], x[0], x[1])
        if op == "l2_loss": return np.sum(x[0] * x[0]) / 2.0
        if op == "softmax_xent":
            z = x[0] - x[0].max(axis=-1, keepdims=True)
            logp = z - np.log(np.exp(z).sum(axis=-1, keepdims=True))
            self.aux[id(t)] = np.exp(logp)
            return -(x[1] * logp).sum(axis=-1)
        if op == "argmax": return np.argmax(x[0], axis=a["axis"])
        if op == "equal": return x[0] == x[1]
        if op == "cast": return x[0].astype(F64) if a["dtype"] in ("float", "float32") else x[0].astype(a["dtype"])
        if op == "assign":
            a["ref"].value = np.array(x[0], F64)
            return a["ref"].value
        if op == "truncated_normal":
            rng, shape = self.graph.rng(), a["shape"]
            s = rng.standard_normal(shape)
            bad = np.abs(s) > 2.0
            while bad.any():
                s[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(s) > 2.0
            return x[0] + x[1] * s
        if op == "random_uniform":
            return self.graph.rng().uniform(float(x[0]), float(x[1]), size=a["shape"])
        raise NotImplementedError(op)

    # ---- reverse mode ----
    def gradients(self, loss, variables):
        order, seen = [], set()
        stack = [(loss, False)]
        while stack:
            t, done = stack.pop()
            if done:
                order.append(t)
                continue
            if id(t) in seen or id(t) not in self.val:
                continue
            seen.add(id(t))
            stack.append((t, True))
            for i in self.active_inputs(t):
                stack.append((i, False))
        grads = {id(loss): np.ones_like(np.asarray(self.val[id(loss)], F64))}
        for t in reversed(order):
            g = grads.get(id(t))
            if g is None or t.op in ("variable", "placeholder", "const"):
                continue
            ins = self.active_inputs(t)
            for i, gi in zip(ins, self._vjp(t, g, [self.val.get(id(i)) for i in ins])):
                if gi is not None:
                    grads[id(i)] = grads[id(i)] + gi if id(i) in grads else gi
        return [grads.get(id(v), np.zeros_like(v.value)) for v in variables]

    def _vjp(self, t, g, x):
        op, a = t.op, t.attrs
        y = self.val[id(t)]
        if op == "cond": return [g]
        if op == "add": return [_unbroadcast(g, x[0].shape), _unbroadcast(g, x[1].shape)]
        if op == "sub": return [_unbroadcast(g, x[0].shape), _unbroadcast(-g, x[1].shape)]
        if op == "mul": return [_unbroadcast(g * x[1], x[0].shape), _unbroadcast(g * x[0], x[1].shape)]
        if op == "div": return [_unbroadcast(g / x[1], x[0].shape), _unbroadcast(-g * x[0] / (x[1] * x[1]), x[1].shape)]
        if op == "maximum":      # the gradient goes to x where x >= y, else to y (TF's MaximumGrad)
            m = x[0] >= x[1]
            return [_unbroadcast(g * m, x[0].shape), _unbroadcast(g * ~m, x[1].shape)]
        if op == "minimum":      # to x where x <= y, else to y
            m = x[0] <= x[1]
            return [_unbroadcast(g * m, x[0].shape), _unbroadcast(g * ~m, x[1].shape)]
        if op == "neg": return [-g]
        if op == "sqrt": return [g * 0.5 / y]
        if op == "rsqrt": return [g * (-0.5) * y ** 3]
        if op == "square": return [g * 2.0 * x[0]]
        if op == "tanh": return [g * (1.0 - y * y)]
        if op == "stop_gradient": return [None]
        if op == "identity": return [g]
        if op == "relu": return [g * (y > 0)]
        if op == "leaky_relu": return [g * np.where(x[0] > 0, 1.0, a["alpha"])]
        if op == "softmax": return [(g - (g * y).sum(axis=-1, keepdims=True)) * y]
        if op in ("reduce_mean", "reduce_sum"):
            axes = list(range(x[0].ndim)) if a["axes"] is None else [ax % x[0].ndim for ax in a["axes"]]
            gg = g if a["keepdims"] else np.expand_dims(g, tuple(sorted(axes)))
            n = np.prod([x[0].shape[ax] for ax in axes]) if op == "reduce_mean" else 1.0
            return [np.broadcast_to(gg, x[0].shape) / n]
        if op == "squeeze": return [g.reshape(x[0].shape)]
        if op == "bias_add": return [g, g.reshape(-1, g.shape[-1]).sum(axis=0)]
        if op == "conv1d": return list(_conv_backward(x[0], x[1], a["dilation"], g))
        if op == "matmul": return [g @ x[1].T, x[0].T @ g]
        if op == "dropout":
            mask, keep = self.aux[id(t)]
            return [g / keep * mask, None]
        if op == "concat":
            out, pos = [], 0
            for xi in x:
                idx = [slice(None)] * g.ndim
                idx[a["axis"]] = slice(pos, pos + xi.shape[a["axis"]])
                out.append(g[tuple(idx)])
                pos += xi.shape[a["axis"]]
            return out
        if op == "slice":
            gx = np.zeros_like(x[0])
            idx = [slice(None)] * gx.ndim
            idx[a["axis"]] = slice(a["start"], a["start"] + a["size"])
            gx[tuple(idx)] = g
            return [gx]
        if op == "einsum":
            sa, sb = a["subs"]
            return [np.einsum("%s,%s->%s" % (a["out"], sb, sa), g, x[1]), np.einsum("%s,%s->%s" % (a["out"], sa, sb), g, x[0])]
        if op == "l2_loss": return [g * x[0]]
        if op == "softmax_xent": return [g[:, None] * (self.aux[id(t)] - x[1]), None]
        if op in ("argmax", "equal", "cast"): return [None] * len(x)
        if op == "assign": return [None]
        raise NotImplementedError("gradient of " + op)

    def _adam(self, t):
        a = t.attrs
        loss, lr = t.inputs
        self.get(loss)
        lr = float(self.get(lr))
        grads = self.gradients(loss, a["vars"])
        self.aux["gradients"] = dict((v.name, g) for v, g in zip(a["vars"], grads))
        p1, p2 = a["powers"]
        lr_t = lr * np.sqrt(1.0 - p2.value) / (1.0 - p1.value)
        for v, g, (m, s) in zip(a["vars"], grads, a["slots"]):
            m.value = a["b1"] * m.value + (1.0 - a["b1"]) * g
            s.value = a["b2"] * s.value + (1.0 - a["b2"]) * g * g
            v.value = v.value - lr_t * m.value / (np.sqrt(s.value) + a["eps"])
        p1.value = p1.value * a["b1"]
        p2.value = p2.value * a["b2"]
        return None


class Session(object):
    last_run = None         # the _Run of the most recent Session.run (a generator reads gradients / intermediates from it)

    def __init__(self, target="", graph=None, config=None):
        self.graph = graph if graph is not None else get_default_graph()        # tf.Session binds the default graph when it is made

    def __enter__(self):                # "with tf.Session(graph=g):" also makes g the default graph (BaseSession.__enter__)
        _GRAPHS.append(self.graph)
        return self

    def __exit__(self, *exc):
        _GRAPHS.pop()
        return False

    def close(self):
        pass

    def run(self, fetches, feed_dict=None):
        if sys.getrecursionlimit() < 20000:
            sys.setrecursionlimit(20000)
        run = _Run(self.graph, feed_dict)
        Session.last_run = run
        many = isinstance(fetches, (list, tuple))
        # training ops last: the values fetched beside them are those of the forward pass the gradients came from
        todo = sorted(range(len(fetches)), key=lambda i: fetches[i].op == "minimize") if many else None
        if not many:
            out = self._out(fetches, run.get(fetches))
        else:
            out = [None] * len(fetches)
            for i in todo:
                out[i] = self._out(fetches[i], run.get(fetches[i]))
        if RUN_HOOK[0] is not None:
            RUN_HOOK[0](fetches, run)
        return out

    @staticmethod
    def _out(t, v):
        if v is None or isinstance(t, Operation):
            return None
        v = np.asarray(v)
        if v.dtype == F64 and not FETCH_FLOAT64[0]:
            v = v.astype(np.float32)
        return v[()] if v.ndim == 0 else v


class ConfigProto(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


# ------------------------------------------------------------------------------------------------------------------------
# tf.train.Saver / import_meta_graph: graph -> model.meta (pickle), values -> model.data-00000-of-00001 (npz by variable name)
# ------------------------------------------------------------------------------------------------------------------------
DATA_SUFFIX = ".data-00000-of-00001"


class Saver(object):
    def __init__(self, var_list=None):
        self.graph = get_default_graph()

    def save(self, sess, save_path):
        g = sess.graph
        lim = sys.getrecursionlimit()
        sys.setrecursionlimit(100000)
        try:
            with open(save_path + ".meta", "wb") as f:
                pickle.dump(g, f, protocol=4)
        finally:
            sys.setrecursionlimit(lim)
        values = dict((v.name, v.value) for v in g.get_collection(GraphKeys.GLOBAL_VARIABLES))
        with open(save_path + DATA_SUFFIX, "wb") as f:
            np.savez(f, **values)
        with open(save_path + ".index", "wt") as f:
            f.write("\n".join(sorted(values)) + "\n")
        with open(os.path.join(os.path.dirname(save_path), "checkpoint"), "wt") as f:
            f.write('model_checkpoint_path: "%s"\n' % os.path.basename(save_path))
        return save_path

    def restore(self, sess, save_path):
        with np.load(save_path + DATA_SUFFIX) as z:
            for v in sess.graph.get_collection(GraphKeys.GLOBAL_VARIABLES):
                v.value = np.array(z[v.name], F64)


def import_meta_graph(meta_path):
    lim = sys.getrecursionlimit()
    sys.setrecursionlimit(100000)
    try:
        with open(meta_path, "rb") as f:
            g = pickle.load(f)
    finally:
        sys.setrecursionlimit(lim)
    get_default_graph().adopt(g)
    return Saver()


def read_checkpoint(model_prefix):
    """{variable name: float64 array} of a checkpoint written by Saver.save (generator-side helper, not TF API)."""
    with np.load(model_prefix + DATA_SUFFIX) as z:
        return dict((k, z[k]) for k in z.files)


def write_checkpoint_values(model_prefix, values):
    """Overwrite the values of named variables in a checkpoint written by Saver.save (generator-side helper: this is how a
    fixture's weights enter a model the reference's build_model created; names and shapes must already exist there)."""
    cur = read_checkpoint(model_prefix)
    for k, v in values.items():
        assert k in cur, "the reference's graph has no variable %s" % k
        assert tuple(cur[k].shape) == tuple(np.shape(v)), (k, cur[k].shape, np.shape(v))
        cur[k] = np.asarray(v, F64)
    with open(model_prefix + DATA_SUFFIX, "wb") as f:
        np.savez(f, **cur)


# ------------------------------------------------------------------------------------------------------------------------
def install():
    """Register this module as ``tensorflow`` (+ the two sub-module paths tf_block.py:2 imports)."""
    me = sys.modules[__name__]
    tf = types.ModuleType("tensorflow")
    for k in ("placeholder", "constant", "zeros", "ones", "truncated_normal", "random_uniform", "constant_initializer", "get_variable",
              "Variable", "global_variables_initializer", "add", "subtract", "multiply", "divide", "maximum", "minimum", "negative", "sqrt",
              "rsqrt", "square", "stop_gradient", "identity", "reduce_mean", "reduce_sum", "squeeze", "concat", "split", "einsum", "argmax",
              "equal", "cast", "assign", "cond", "matmul", "Session", "ConfigProto", "Graph", "GraphKeys", "get_default_graph",
              "reset_default_graph", "set_random_seed", "get_collection", "name_scope", "variable_scope", "control_dependencies", "float32"):
        setattr(tf, k, getattr(me, k))
    tf.bool = bool_
    tf.nn = types.SimpleNamespace(conv1d=conv1d, convolution=convolution, bias_add=bias_add, relu=relu, leaky_relu=leaky_relu,
                                  moments=moments, batch_normalization=batch_normalization, dropout=dropout, xw_plus_b=xw_plus_b,
                                  l2_loss=l2_loss, softmax_cross_entropy_with_logits=softmax_cross_entropy_with_logits, tanh=tanh,
                                  softmax=softmax)
    tf.train = types.SimpleNamespace(AdamOptimizer=AdamOptimizer, Saver=Saver, import_meta_graph=import_meta_graph)
    tf.contrib = types.SimpleNamespace(layers=types.SimpleNamespace(xavier_initializer=xavier_initializer))
    tf.numpy_tf1 = me
    tfp = types.ModuleType("tensorflow.python")
    tfpf = types.ModuleType("tensorflow.python.framework")
    tfpf.ops = types.SimpleNamespace(name_scope=name_scope)
    tfp.framework = tfpf
    tf.python = tfp
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.python"] = tfp
    sys.modules["tensorflow.python.framework"] = tfpf
    return tf
