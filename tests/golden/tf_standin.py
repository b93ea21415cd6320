"""A numpy evaluator for the TensorFlow-1.x calls of the reference's GENERATION path (TEST TOOLING, build container only).

Why this exists.  The reference (bfs18/nsynth_wavenet) is TensorFlow-1.x Python and TensorFlow is not installed here, so
its float path could only be checked against a restatement
(oracle/wavenet_np.py).  This module lets the reference's OWN files -- wavenet/masked.py, wavenet/wavenet.py,
wavenet/parallel_wavenet.py, wavenet/loss_func.py, wavenet/fastgen.py, wavenet/parallelgen.py, auxilaries/utils.py, imported
from /root/reference, unmodified -- build and run their graphs: `import tensorflow as tf` resolves to this module, which
implements the ~70 primitives those files call (graph / session / placeholder / variable scopes / Saver.restore, conv2d,
conv2d_transpose, pad, slice, reshape, FIFOQueue, random_uniform, ...) in numpy, following TensorFlow's documented op
semantics.  tests/golden/make_ref_float.py uses it to produce tests/golden/ref_float_*.npz.

What the resulting vectors pin, and what they do not.  PINNED: everything the reference's own code decides -- variable
names and scopes, checkpoint (EMA) names, time_to_batch / batch_to_time dilation arithmetic, causal padding, tap order of
the kernels, centre crop of the conditioning, order of the sigmoid / tanh halves, residual / skip wiring, flow head,
mean_tot / scale_tot recursion and clips, the queue discipline of the incremental sampler, the sampler formulas, the
quantiser, the driver loops of fastgen.synthesis / parallelgen.synthesis down to the wav files they write.  NOT PINNED:
TensorFlow's kernels themselves (this file restates their semantics: SAME-padding arithmetic of conv2d_transpose,
nearest-neighbour resize, l2_normalize epsilon, softplus), float32 rounding order inside a TF kernel, and TF's random
number generators (randoms are injected: `random_source`).  It is NOT a TensorFlow run and the fixtures say so.

Nothing here is imported by the product, the oracle, bench.py or any test that runs on the GPU box; the tests read the
.npz files only.  Evaluation is deferred like TF-1.x: ops build nodes, `Session.run` evaluates them.  Static shapes come
from evaluating every node once on zero-filled placeholders / variables at construction time.
"""
import contextlib
import sys
import types

import numpy as np

# arithmetic type that `tf.float32` maps to: np.float32 = what TF computes in, np.float64 = master copy
FLOAT = [np.float64]


def set_float(dt):
    FLOAT[0] = dt


class DType(object):
    def __init__(self, name, npdt):
        self.name, self._np = name, npdt

    @property
    def np(self):
        return FLOAT[0] if self._np is None else self._np

    def __repr__(self):
        return 'tf.' + self.name


float32 = DType('float32', None)
float64 = DType('float64', np.float64)
int32 = DType('int32', np.int32)
int64 = DType('int64', np.int64)
int8 = DType('int8', np.int8)
bool_ = DType('bool', np.bool_)
AUTO_REUSE = object()


class TensorShape(object):
    def __init__(self, dims):
        self._dims = [None if d is None else int(d) for d in dims]

    def as_list(self):
        return list(self._dims)

    @property
    def ndims(self):
        return len(self._dims)

    def __len__(self):
        return len(self._dims)

    def __getitem__(self, i):
        return self._dims[i]

    def __iter__(self):
        return iter(self._dims)


# ------------------------------------------------------------------------------------------------------------------
# graph
# ------------------------------------------------------------------------------------------------------------------
class Graph(object):
    def __init__(self):
        self.vars = {}            # full name -> Variable, creation order
        self.scope = []
        self.n_random = 0
        self.nodes = []

    @contextlib.contextmanager
    def as_default(self):
        _GRAPHS.append(self)
        try:
            yield self
        finally:
            _GRAPHS.pop()


_GRAPHS = [Graph()]


def get_default_graph():
    return _GRAPHS[-1]


def reset_default_graph():
    _GRAPHS[:] = [Graph()]


class Tensor(object):
    __array_priority__ = 1000.0
    __array_ufunc__ = None

    def __init__(self, fn, inputs, sample, kind='op', name=None):
        self.fn, self.inputs, self.kind, self.name = fn, list(inputs), kind, name
        self.sample = np.asarray(sample) if sample is not None else None
        self.const = kind == 'const' or (kind == 'op' and all(i.const for i in self.inputs))
        self.graph = get_default_graph()
        self.graph.nodes.append(self)

    # static shape
    def get_shape(self):
        return TensorShape(self.sample.shape)

    @property
    def shape(self):
        return self.get_shape()

    def set_shape(self, shape):
        have = list(self.sample.shape)
        want = list(shape.as_list() if isinstance(shape, TensorShape) else shape)
        assert len(have) == len(want), (have, want)
        for h, w in zip(have, want):
            assert w is None or int(w) == w and int(w) == h, ('set_shape mismatch', have, want)

    @property
    def dtype(self):
        return self.sample.dtype

    def __repr__(self):
        return '<standin Tensor {} {} {}>'.format(self.kind, self.name or '', None if self.sample is None else self.sample.shape)

    # arithmetic (python / numpy scalars take the tensor's dtype, as TF converts them)
    def __add__(self, o): return _bin(np.add, self, o)
    def __radd__(self, o): return _bin(np.add, o, self)
    def __sub__(self, o): return _bin(np.subtract, self, o)
    def __rsub__(self, o): return _bin(np.subtract, o, self)
    def __mul__(self, o): return _bin(np.multiply, self, o)
    def __rmul__(self, o): return _bin(np.multiply, o, self)
    def __truediv__(self, o): return _bin(np.true_divide, self, o)
    def __rtruediv__(self, o): return _bin(np.true_divide, o, self)
    def __pow__(self, o): return _bin(np.power, self, o)
    def __rpow__(self, o): return _bin(np.power, o, self)
    def __neg__(self): return _op(np.negative, self)
    def __lt__(self, o): return _bin(np.less, self, o)
    def __gt__(self, o): return _bin(np.greater, self, o)
    def __le__(self, o): return _bin(np.less_equal, self, o)
    def __ge__(self, o): return _bin(np.greater_equal, self, o)
    __hash__ = object.__hash__

    def __getitem__(self, idx):
        return _op(lambda a: a[idx], self)

    def __iter__(self):
        raise TypeError('standin Tensor is not iterable')


def _is_float(a):
    return np.issubdtype(np.asarray(a).dtype, np.floating)


def constant(value, dtype=None, shape=None, name=None):
    a = np.asarray(value, dtype=None if dtype is None else dtype.np)
    if dtype is None and _is_float(a):
        a = a.astype(FLOAT[0])
    if shape is not None:
        a = np.broadcast_to(a, shape).copy()
    return Tensor(None, [], a, kind='const', name=name)


def convert_to_tensor(x, like=None):
    if isinstance(x, Tensor):
        return x
    a = np.asarray(x)
    if like is not None and _is_float(like.sample) and a.dtype.kind in 'fiu':
        a = a.astype(like.sample.dtype)
    elif like is not None and like.sample.dtype.kind in 'iu' and a.dtype.kind in 'iu':
        a = a.astype(like.sample.dtype)
    elif a.dtype.kind == 'f':
        a = a.astype(FLOAT[0])
    return Tensor(None, [], a, kind='const')


def _op(fn, *tensors, **kw):
    tensors = [convert_to_tensor(t) for t in tensors]
    with np.errstate(all='ignore'):
        sample = fn(*[t.sample for t in tensors])
    return Tensor(fn, tensors, sample, name=kw.get('name'))


def _bin(ufunc, a, b):
    if isinstance(a, Tensor):
        b = convert_to_tensor(b, like=a)
    else:
        a = convert_to_tensor(a, like=b)
    return _op(lambda x, y: ufunc(x, y), a, b)


# ------------------------------------------------------------------------------------------------------------------
# variables, scopes, placeholders
# ------------------------------------------------------------------------------------------------------------------
class Variable(Tensor):
    def __init__(self, full, shape, dtype):
        Tensor.__init__(self, None, [], np.zeros(shape, dtype), kind='var', name=full + ':0')
        self.full = full

    def initialized_value(self):
        return self

    def assign(self, *a, **k):
        raise NotImplementedError('data-dependent initialisation (training path) is outside the stand-in')

    assign_add = assign


@contextlib.contextmanager
def variable_scope(name, reuse=None, **kw):
    g = get_default_graph()
    g.scope.append(name)
    try:
        yield
    finally:
        g.scope.pop()


def get_variable(name, shape=None, dtype=None, initializer=None, **kw):
    g = get_default_graph()
    full = '/'.join(g.scope + [name])
    if full in g.vars:
        v = g.vars[full]
        if shape is not None:
            assert list(v.sample.shape) == [int(s) for s in shape], (full, v.sample.shape, shape)
        return v
    if shape is None:
        raise ValueError('variable {} does not exist'.format(full))
    v = Variable(full, [int(s) for s in shape], (dtype or float32).np)
    g.vars[full] = v
    return v


def trainable_variables():
    return list(get_default_graph().vars.values())


def global_variables():
    return trainable_variables()


def placeholder(dtype, shape=None, name=None):
    return Tensor(None, [], np.zeros([int(s) for s in shape], dtype.np), kind='placeholder', name=name)


def random_normal_initializer(*a, **k):
    return ('random_normal_initializer', a, k)


def constant_initializer(*a, **k):
    return ('constant_initializer', a, k)


# ------------------------------------------------------------------------------------------------------------------
# randoms: injected.  random_source(index, kind, shape, run_no, lo, hi) -> array of FINAL values; index = creation order
# of the random node in its graph; kind 'uniform' (values in [lo, hi)) or 'normal' (standard normal; lo = hi = None)
# ------------------------------------------------------------------------------------------------------------------
class RandomSource(object):
    """Default source: a seeded RandomState, new values at every Session.run; everything drawn is logged."""

    def __init__(self, seed=0):
        self.rs = np.random.RandomState(seed)
        self.log = []

    def __call__(self, index, kind, shape, run_no, lo, hi):
        v = self.rs.uniform(lo, hi, shape) if kind == 'uniform' else self.rs.standard_normal(shape)
        self.log.append((run_no, index, kind, v))
        return v


random_source = [RandomSource(0)]


def set_random_source(src):
    random_source[0] = src


def _random_node(kind, shape, lo=None, hi=None):
    g = get_default_graph()
    idx = g.n_random
    g.n_random += 1
    shape = [int(s) for s in (shape.sample if isinstance(shape, Tensor) else shape)]
    mid = 0.0 if kind == 'normal' else 0.5 * (lo + hi)
    t = Tensor(None, [], np.full(shape, mid, FLOAT[0]), kind='random')
    t.rand = (idx, kind, shape, lo, hi)
    return t


def random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None):
    return _random_node('uniform', shape, float(minval), 1.0 if maxval is None else float(maxval))


def set_random_seed(seed):
    pass


class Normal(object):
    """tf.contrib.distributions.Normal: sample() = loc + scale * n, n ~ N(0, 1) from the random source."""

    def __init__(self, loc, scale, **kw):
        self.loc, self.scale = convert_to_tensor(loc), convert_to_tensor(scale)

    def log_prob(self, value, name=None):
        # tf Normal._log_prob: -0.5 * ((x - loc) / scale)^2 - (0.5 * log(2 pi) + log(scale))
        z = (convert_to_tensor(value, like=self.loc) - self.loc) / self.scale
        return -0.5 * z * z - (0.5 * np.log(2.0 * np.pi) + log(self.scale))

    def sample(self, sample_shape=(), seed=None, name=None):
        base = np.broadcast(self.loc.sample, self.scale.sample).shape
        shape = list(sample_shape) + list(base)
        return self.loc + self.scale * _random_node('normal', shape)


class Categorical(object):
    """tf.distributions.Categorical(logits).sample(n).  TF's multinomial kernel and RNG cannot be reproduced; the draw
    here is by inverse CDF from ONE uniform per row (first index whose running softmax mass exceeds u * total) -- the same
    distribution, and the definition the oracle and the device sampler use for the 'ce' head."""

    def __init__(self, logits=None, **kw):
        self.logits = logits

    def sample(self, n=(), seed=None, name=None):
        n = int(n) if not isinstance(n, (tuple, list)) else int(np.prod(n or [1]))
        lead = list(self.logits.sample.shape[:-1])
        u = _random_node('uniform', [n] + lead, 0.0, 1.0)

        def draw(lg, uu):
            e = np.exp(lg - lg.max(axis=-1, keepdims=True))
            cdf = np.cumsum(e, axis=-1)
            thr = uu[..., None] * cdf[..., -1:][None]
            k = (cdf[None] <= thr).sum(axis=-1)
            return np.minimum(k, lg.shape[-1] - 1).astype(np.int32)
        return _op(draw, self.logits, u)


class Mixture(object):
    def __init__(self, *a, **k):
        raise NotImplementedError('Mixture (training path) is outside the stand-in')


# ------------------------------------------------------------------------------------------------------------------
# queues
# ------------------------------------------------------------------------------------------------------------------
class FIFOQueue(object):
    def __init__(self, capacity, dtypes, shapes=None, **kw):
        self.capacity, self.shape, self.dt = int(capacity), tuple(int(s) for s in shapes), dtypes.np
        self.items = []

    def _enq(self, t, many):
        t = convert_to_tensor(t)
        node = Tensor(None, [t], None, kind='enqueue')
        node.queue, node.many = self, many
        node.const = False
        return node

    def enqueue(self, t, name=None):
        return self._enq(t, False)

    def enqueue_many(self, t, name=None):
        return self._enq(t, True)

    def dequeue(self, name=None):
        node = Tensor(None, [], np.zeros(self.shape, self.dt), kind='dequeue')
        node.queue = self
        return node


# ------------------------------------------------------------------------------------------------------------------
# session
# ------------------------------------------------------------------------------------------------------------------
class _NS(object):
    pass


def ConfigProto(**kw):
    c = _NS()
    c.gpu_options = _NS()
    c.__dict__.update(kw)
    return c


class Session(object):
    def __init__(self, config=None, graph=None, **kw):
        self.graph = graph or get_default_graph()
        self.values = {}         # variable full name -> array
        self.run_no = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass

    def _leaf(self, n, feed):
        if n.kind == 'const':
            return n.sample
        if n.kind == 'placeholder':
            if n not in feed:
                raise ValueError('placeholder {} not fed'.format(n.name))
            v = np.asarray(feed[n]).astype(n.sample.dtype)
            assert v.shape == n.sample.shape, (n.name, v.shape, n.sample.shape)
            return v
        if n.kind == 'var':
            if n.full not in self.values:
                raise ValueError('variable {} is uninitialised (not restored)'.format(n.full))
            return self.values[n.full]
        if n.kind == 'random':
            idx, kind, shape, lo, hi = n.rand
            v = np.asarray(random_source[0](idx, kind, shape, self.run_no, lo, hi), n.sample.dtype)
            assert list(v.shape) == shape, (v.shape, shape)
            return v
        if n.kind == 'dequeue':
            if not n.queue.items:
                raise RuntimeError('dequeue from an empty queue')
            return n.queue.items.pop(0)
        raise AssertionError(n.kind)

    def _value(self, t, memo, feed):
        stack = [t]
        while stack:
            n = stack[-1]
            if id(n) in memo:
                stack.pop()
                continue
            if n.kind == 'enqueue':
                pend = [i for i in n.inputs if id(i) not in memo]
                if pend:
                    stack.extend(pend)
                    continue
                v = memo[id(n.inputs[0])]
                items = list(v) if n.many else [v]
                for it in items:
                    assert np.shape(it) == n.queue.shape, (np.shape(it), n.queue.shape)
                    assert len(n.queue.items) < n.queue.capacity, 'enqueue on a full queue would block'
                    n.queue.items.append(np.array(it, n.queue.dt))
                memo[id(n)] = None
                stack.pop()
                continue
            if n.kind != 'op':
                memo[id(n)] = self._leaf(n, feed)
                stack.pop()
                continue
            pend = [i for i in n.inputs if id(i) not in memo]
            if pend:
                stack.extend(pend)
                continue
            with np.errstate(all='ignore'):
                memo[id(n)] = np.asarray(n.fn(*[memo[id(i)] for i in n.inputs]))
            stack.pop()
        return memo[id(t)]

    def run(self, fetches, feed_dict=None):
        feed = {} if feed_dict is None else dict(feed_dict)
        memo = {}
        self.run_no += 1
        flat = []

        def collect(f):
            if isinstance(f, Tensor):
                flat.append(f)
            elif isinstance(f, dict):
                for v in f.values():
                    collect(v)
            elif isinstance(f, (list, tuple)):
                for v in f:
                    collect(v)
            elif f is None or isinstance(f, (int, float)):
                pass
            else:
                raise TypeError('cannot fetch {!r}'.format(f))
        collect(fetches)
        # a bounded FIFOQueue makes "dequeue, then enqueue" the only order TF can execute within one run
        for f in flat:
            if f.kind != 'enqueue':
                self._value(f, memo, feed)
        for f in flat:
            if f.kind == 'enqueue':
                self._value(f, memo, feed)

        def build(f):
            if isinstance(f, Tensor):
                return memo[id(f)]
            if isinstance(f, dict):
                return {k: build(v) for k, v in f.items()}
            if isinstance(f, (list, tuple)):
                return [build(v) for v in f]
            return f
        return build(fetches)


class Saver(object):
    """tf.train.Saver(var_dict, reshape=...).restore(sess, path): `path` is an .npz whose keys are CHECKPOINT names.
    Every requested name must be present (TF raises NotFoundError otherwise); the names requested are recorded."""
    requested = []

    def __init__(self, var_list=None, reshape=False, **kw):
        if var_list is None:
            var_list = {v.name[:-2]: v for v in trainable_variables()}
        elif not isinstance(var_list, dict):
            var_list = {v.name[:-2]: v for v in var_list}
        self.var_list, self.reshape = var_list, reshape

    def restore(self, sess, save_path):
        ck = np.load(save_path)
        for key, var in self.var_list.items():
            if key not in ck.files:
                raise KeyError('Key {} not found in checkpoint'.format(key))
            a = np.asarray(ck[key])
            if tuple(a.shape) != tuple(var.sample.shape):
                if not (self.reshape and a.size == var.sample.size):
                    raise ValueError('shape of {} in the checkpoint {} != {}'.format(key, a.shape, var.sample.shape))
                a = a.reshape(var.sample.shape)
            sess.values[var.full] = a.astype(var.sample.dtype)
        Saver.requested.append(sorted(self.var_list.keys()))


# ------------------------------------------------------------------------------------------------------------------
# ops
# ------------------------------------------------------------------------------------------------------------------
def _axis(a):
    return tuple(a) if isinstance(a, (list, tuple)) else a


def _static(x):
    """python value of a shape-like argument (ints, lists that may hold const Tensors, const Tensors)."""
    if isinstance(x, Tensor):
        assert x.const, 'shape argument must be static'
        return [int(v) for v in np.asarray(x.sample).reshape(-1)]
    if isinstance(x, (list, tuple)):
        return [int(_static(v)[0]) if isinstance(v, Tensor) else int(v) for v in x]
    return int(x)


def pad(x, paddings, mode='CONSTANT', name=None):
    p = [tuple(int(v) for v in r) for r in paddings]
    return _op(lambda a: np.pad(a, p), x)


def slice(x, begin, size, name=None):        # noqa: A001  (TF's own name)
    b, s = _static(begin), _static(size)

    def f(a):
        idx = tuple(np.s_[bi:(a.shape[i] if si == -1 else bi + si)] for i, (bi, si) in enumerate(zip(b, s)))
        return a[idx]
    return _op(f, x)


def stack(values, axis=0, name=None):
    return _op(lambda *a: np.stack(a, axis=axis), *values)


def concat(values, axis, name=None):
    return _op(lambda *a: np.concatenate(a, axis=axis), *values)


def reshape(x, shape, name=None):
    s = _static(shape)
    return _op(lambda a: a.reshape(s), x)


def transpose(x, perm=None, name=None):
    return _op(lambda a: np.transpose(a, perm), x)


def expand_dims(x, axis=None, name=None, dim=None):
    ax = dim if axis is None else axis
    return _op(lambda a: np.expand_dims(a, ax), x)


def squeeze(x, axis=None, name=None, squeeze_dims=None):
    ax = squeeze_dims if axis is None else axis
    return _op(lambda a: np.squeeze(a, axis=_axis(ax)), x)


def tile(x, multiples, name=None):
    m = _static(multiples)
    return _op(lambda a: np.tile(a, m), x)


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    n = num_or_size_splits
    if isinstance(n, int):
        size = value.sample.shape[axis] // n
        assert size * n == value.sample.shape[axis]
        edges = [(i * size, (i + 1) * size) for i in range(n)]
    else:
        c = np.cumsum([0] + list(n))
        edges = list(zip(c[:-1], c[1:]))
    outs = []
    for lo, hi in edges:
        outs.append(_op((lambda lo, hi: lambda a: np.take(a, np.arange(lo, hi), axis=axis))(int(lo), int(hi)), value))
    return outs


def identity(x, name=None):
    return _op(lambda a: a, x)


def zeros(shape, dtype=float32, name=None):
    return constant(np.zeros(_static(shape), dtype.np))


def ones(shape, dtype=float32, name=None):
    return constant(np.ones(_static(shape), dtype.np))


def shape(x, name=None):        # noqa: A001
    return Tensor(None, [], np.asarray(x.sample.shape, np.int32), kind='const')


def cast(x, dtype, name=None):
    return _op(lambda a: np.asarray(a).astype(dtype.np), x)


def _un(f):
    return lambda x, name=None: _op(f, x)


floor = _un(np.floor)
sign = _un(np.sign)
abs = _un(np.abs)                # noqa: A001
log = _un(np.log)
exp = _un(np.exp)
sqrt = _un(np.sqrt)
square = _un(np.square)
tanh = _un(np.tanh)


def sigmoid(x, name=None):
    # 1 / (1 + exp(-x)) evaluated without overflow on either side
    def f(a):
        e = np.exp(-np.abs(a))
        return np.where(a >= 0, 1.0 / (1.0 + e), e / (1.0 + e)).astype(a.dtype)
    return _op(f, x)


def pow(x, y, name=None):        # noqa: A001
    return _bin(np.power, x, y)


def maximum(x, y, name=None):
    return _bin(np.maximum, x, y)


def minimum(x, y, name=None):
    return _bin(np.minimum, x, y)


def squared_difference(x, y, name=None):
    return _bin(lambda a, b: (a - b) * (a - b), x, y)


def equal(x, y, name=None):
    return _bin(np.equal, x, y)


def clip_by_value(x, lo, hi, name=None):
    x = convert_to_tensor(x)
    lo, hi = convert_to_tensor(lo, like=x), convert_to_tensor(hi, like=x)
    return _op(lambda a, l, h: np.minimum(np.maximum(a, l), h), x, lo, hi)


def where(cond, x=None, y=None, name=None):
    x = convert_to_tensor(x)
    y = convert_to_tensor(y, like=x)
    return _op(lambda c, a, b: np.where(c, a, b), cond, x, y)


def _red(f):
    def g(x, axis=None, keepdims=False, name=None, keep_dims=None, reduction_indices=None):
        kd = keepdims if keep_dims is None else keep_dims
        ax = reduction_indices if axis is None else axis
        return _op(lambda a: f(a, axis=_axis(ax), keepdims=kd), x)
    return g


reduce_sum = _red(np.sum)
reduce_mean = _red(np.mean)
reduce_max = _red(np.max)
reduce_min = _red(np.min)


def matmul(a, b, name=None):
    return _op(lambda x, y: x @ y, a, b)


def argmax(x, axis=None, name=None, dimension=None, output_type=int64):
    ax = dimension if axis is None else axis
    return _op(lambda a: np.argmax(a, axis=ax).astype(np.int64), x)


def one_hot(indices, depth, on_value=None, off_value=None, axis=None, dtype=float32, name=None):
    return _op(lambda i: (np.asarray(i)[..., None] == np.arange(depth)).astype(dtype.np), indices)


@contextlib.contextmanager
def control_dependencies(ops):
    yield


@contextlib.contextmanager
def name_scope(*a, **k):
    yield


@contextlib.contextmanager
def device(*a, **k):
    yield


def gradients(*a, **k):
    raise NotImplementedError('training path')


def global_variables_initializer():
    raise NotImplementedError('initialisers are not evaluated; restore the variables from a checkpoint')


# ---- tf.nn -------------------------------------------------------------------------------------------------------
def _conv2d(x, w, padding):
    # NHWC with H == 1, filter [1, K, Cin, Cout], stride 1; cross-correlation (no kernel flip) like TF
    assert x.ndim == 4 and x.shape[1] == 1 and w.shape[0] == 1 and x.shape[3] == w.shape[2], (x.shape, w.shape)
    K = w.shape[1]
    if padding == 'SAME':
        tot = K - 1
        x = np.pad(x, ((0, 0), (0, 0), (tot // 2, tot - tot // 2), (0, 0)))
    else:
        assert padding == 'VALID'
    n = x.shape[2] - K + 1
    y = np.zeros(x.shape[:2] + (n, w.shape[3]), np.result_type(x, w))
    for k in range(K):
        y += x[:, :, k:k + n, :] @ w[0, k]
    return y


def _conv2d_transpose(x, w, out_shape, stride, padding):
    # gradient of conv2d(input [N,1,Lout,Cout] -> [N,1,L,Cin], filter [1,K,Cout,Cin], stride, SAME) w.r.t. its input.
    # forward SAME padding: total = max((L - 1) * stride + K - Lout, 0), left = total // 2;
    # forward o[j] = sum_k in[j * stride + k - left] W[k]  =>  y[i] = sum_{j,k: j*stride + k - left = i} x[j] W[k]^T
    assert padding == 'SAME' and x.shape[1] == 1 and w.shape[0] == 1 and x.shape[3] == w.shape[3]
    N, _, L, _ = x.shape
    K, Cout = w.shape[1], w.shape[2]
    Lout = int(out_shape[2])
    assert -(-Lout // stride) == L and int(out_shape[3]) == Cout and int(out_shape[0]) == N
    tot = max((L - 1) * stride + K - Lout, 0)
    left = tot // 2
    full = np.zeros((N, max((L - 1) * stride + K, left + Lout), Cout), np.result_type(x, w))
    for k in range(K):
        full[:, k:k + (L - 1) * stride + 1:stride, :] += x[:, 0] @ w[0, k].T
    return full[:, left:left + Lout][:, None]


def _softplus(a):
    return (np.maximum(a, 0) + np.log1p(np.exp(-np.abs(a)))).astype(a.dtype)


nn = types.ModuleType('tensorflow.nn')
nn.conv2d = lambda input, filter, strides, padding, name=None, **kw: (          # noqa: A002
    _chk(list(strides) == [1, 1, 1, 1]), _op(lambda a, w: _conv2d(a, w, padding), input, filter))[1]
nn.conv2d_transpose = lambda value, filter, output_shape, strides, padding='SAME', name=None, **kw: (   # noqa: A002
    _chk(strides[0] == 1 and strides[1] == 1 and strides[3] == 1),
    _op(lambda a, w: _conv2d_transpose(a, w, _static(list(output_shape)), int(strides[2]), padding), value, filter))[1]
nn.bias_add = lambda value, bias, name=None, **kw: _op(lambda a, b: a + b, value, bias, name='bias_add')
nn.relu = _un(lambda a: np.maximum(a, 0))
nn.leaky_relu = lambda features, alpha=0.2, name=None: _op(lambda a: np.maximum(a, a * a.dtype.type(alpha)), features)
nn.tanh = tanh
nn.sigmoid = sigmoid
nn.softplus = _un(_softplus)
nn.l2_normalize = lambda x, axis=None, epsilon=1e-12, name=None, dim=None: _op(
    lambda a: a / np.sqrt(np.maximum(np.sum(a * a, axis=_axis(dim if axis is None else axis), keepdims=True), epsilon)), x)


def _moments(*a, **k):
    raise NotImplementedError('data-dependent initialisation (training path)')


nn.moments = _moments


def _sparse_xent(_sentinel=None, labels=None, logits=None, name=None):
    def f(lg, lb):
        m = lg.max(axis=-1, keepdims=True)
        ls = lg - m - np.log(np.sum(np.exp(lg - m), axis=-1, keepdims=True))
        return -np.take_along_axis(ls, lb[..., None].astype(np.int64), axis=-1)[..., 0]
    return _op(f, logits, labels)


nn.sparse_softmax_cross_entropy_with_logits = _sparse_xent


def _chk(ok):
    assert ok
    return None


# ---- tf.image / tf.layers / summaries / logging -------------------------------------------------------------------
image = types.ModuleType('tensorflow.image')


def _resize_nn(images, size, align_corners=False, name=None):
    h, w = _static(list(size))

    def f(a):
        assert not align_corners
        ih, iw = a.shape[1], a.shape[2]
        yi = np.minimum((np.arange(h) * (ih / h)).astype(np.int64), ih - 1)     # floor(i * in / out)
        xi = np.minimum((np.arange(w) * (iw / w)).astype(np.int64), iw - 1)
        return a[:, yi][:, :, xi]
    return _op(f, images)


image.resize_nearest_neighbor = _resize_nn

layers = types.ModuleType('tensorflow.layers')


def _dropout(inputs, rate=0.5, noise_shape=None, seed=None, training=False, name=None):
    assert training is False, 'dropout in training mode is outside the generation path'
    return identity(inputs)


layers.dropout = _dropout
layers.conv2d_transpose = _moments

summary = types.ModuleType('tensorflow.summary')
summary.scalar = lambda *a, **k: None
summary.histogram = lambda *a, **k: None
summary.merge_all = lambda *a, **k: None

logging = types.ModuleType('tensorflow.logging')
logging.info = lambda *a, **k: None
logging.warning = lambda *a, **k: None
logging.set_verbosity = lambda *a, **k: None
logging.INFO = 20

train = types.ModuleType('tensorflow.train')
train.Saver = Saver

distributions = types.ModuleType('tensorflow.distributions')
distributions.Categorical = Categorical
distributions.Normal = Normal


class HParams(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def values(self):
        return dict(self.__dict__)


contrib = types.ModuleType('tensorflow.contrib')
contrib.distributions = types.ModuleType('tensorflow.contrib.distributions')
contrib.distributions.Normal = Normal
contrib.distributions.Categorical = Categorical
contrib.distributions.Mixture = Mixture
contrib.training = types.ModuleType('tensorflow.contrib.training')
contrib.training.HParams = HParams
contrib.slim = types.ModuleType('tensorflow.contrib.slim')
contrib.signal = types.ModuleType('tensorflow.contrib.signal')


def install():
    """Register this module as `tensorflow` (and its sub-modules) in sys.modules.  Returns the names it set."""
    me = sys.modules[__name__]
    names = {
        'tensorflow': me, 'tensorflow.nn': nn, 'tensorflow.image': image, 'tensorflow.layers': layers,
        'tensorflow.summary': summary, 'tensorflow.logging': logging, 'tensorflow.train': train,
        'tensorflow.distributions': distributions, 'tensorflow.contrib': contrib,
        'tensorflow.contrib.distributions': contrib.distributions, 'tensorflow.contrib.training': contrib.training,
        'tensorflow.contrib.slim': contrib.slim, 'tensorflow.contrib.signal': contrib.signal,
    }
    for k, v in names.items():
        sys.modules[k] = v
    return list(names)


class _Unsupported(object):
    """Anything of TensorFlow this file does not implement: may be named and called at IMPORT time (the reference's
    training-side modules declare TFRecord features, HParams ... at module level); its result is another inert object
    that supports nothing, so reaching one on the evaluated path fails at the first use."""

    def __init__(self, name):
        self.__dict__['_name'] = name

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Unsupported(self._name + '.' + name)

    def __call__(self, *a, **k):
        return _Unsupported(self._name + '()')

    def __repr__(self):
        return '<tensorflow stand-in: unsupported {}>'.format(self._name)


def __getattr__(name):
    if name.startswith('__'):
        raise AttributeError(name)
    return _Unsupported('tf.' + name)
