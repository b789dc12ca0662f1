"""TEST INFRASTRUCTURE — numpy restatement of the reference autograd (core/tensor.py, core/ops.py).

Faithful where it matters for parity and for the CPU baseline:
  * values are whatever numpy promotes to; every node that requires grad owns a float64 zero gradient
    (core/tensor.py:24-25,170-171) so backward math is float64 and parameters become float64 after the
    first `+=`;
  * backward recurses once per dependency EDGE, accumulating into `.grad` on the way (core/tensor.py:
    157-168) — no topological sort, hence the 4x traversal of the MLP under the softmax loss;
  * vjp formulas, tie rules and the un-broadcast rule follow core/ops.py line by line (cited per op).
Written table-style (one `_node` helper + a shared `_unbroadcast`) rather than as a transcription.
"""

import numpy as np


class RefTensor(object):

    def __init__(self, values, requires_grad=False, dependency=None, dtype=None):
        self._values = np.asarray(values, dtype)          # core/tensor.py:20
        self.grad = None
        self.requires_grad = requires_grad
        if requires_grad:
            self.zero_grad()
        self.dependency = dependency or []

    values = property(lambda self: self._values)

    @values.setter
    def values(self, new):                                 # core/tensor.py:35-38
        self._values = np.asarray(new)
        self.grad = None

    shape = property(lambda self: self._values.shape)

    def zero_grad(self):                                   # core/tensor.py:170-171 (float64!)
        self.grad = np.zeros(self.shape)

    def backward(self, grad=None):                         # core/tensor.py:157-168
        assert self.requires_grad, "Call backward() on a non-requires-grad tensor."
        grad = np.array(1.0 if grad is None else grad)
        self.grad += grad
        for dep in self.dependency:
            dep["tensor"].backward(dep["grad_fn"](grad))

    def __len__(self):
        return len(self._values)

    # comparisons return raw bool arrays (core/tensor.py:48-58)
    def __gt__(self, o): return self.values > _t(o).values
    def __lt__(self, o): return self.values < _t(o).values
    def __ge__(self, o): return self.values >= _t(o).values
    def __le__(self, o): return self.values <= _t(o).values

    def __add__(self, o): return add(self, _t(o))
    def __radd__(self, o): return add(_t(o), self)
    def __sub__(self, o): return sub(self, _t(o))
    def __rsub__(self, o): return sub(_t(o), self)
    def __mul__(self, o): return mul(self, _t(o))
    def __rmul__(self, o): return mul(_t(o), self)
    def __truediv__(self, o): return div(self, _t(o))
    def __rtruediv__(self, o): return div(_t(o), self)
    def __pow__(self, o): return power(self, _t(o))
    def __matmul__(self, o): return dot(self, _t(o))
    def __neg__(self): return neg(self)
    def __getitem__(self, key): return getitem(self, key)

    def _replace(self, new):                               # non-autograd in-place forms (:66-68 ...)
        self.values = new
        return self

    def __iadd__(self, o): return self._replace(self.values + _t(o).values)
    def __isub__(self, o): return self._replace(self.values - _t(o).values)
    def __imul__(self, o): return self._replace(self.values * _t(o).values)
    def __itruediv__(self, o): return self._replace(self.values / _t(o).values)

    def sum(self, axis=None): return rsum(self, axis)
    def max(self, axis=None): return rmax(self, axis)
    def min(self, axis=None): return rmin(self, axis)
    def transpose(self, axes=None): return transpose(self, axes)
    T = property(lambda self: transpose(self, None))
    def log(self): return log(self)
    def reshape(self, newshape): return reshape(self, newshape)
    def flatten(self): return flatten(self)
    def clip(self, min=None, max=None): return clip(self, min, max)


def _t(obj):
    return obj if isinstance(obj, RefTensor) else RefTensor(obj)


def _node(values, *parents):
    """core/ops.py:12-29 — edges only towards inputs that require grad; output class = first input's."""
    edges = [dict(tensor=t, grad_fn=fn) for t, fn in parents if t.requires_grad]
    return parents[0][0].__class__(values, bool(edges), edges)


def _unbroadcast(grad, like):
    """core/ops.py:41-46"""
    for _ in range(grad.ndim - like.values.ndim):
        grad = grad.sum(axis=0)
    for i, dim in enumerate(like.shape):
        if dim == 1:
            grad = grad.sum(axis=i, keepdims=True)
    return grad


def add(a, b):      # core/ops.py:32-58
    return _node(a.values + b.values, (a, lambda g: _unbroadcast(g, a)), (b, lambda g: _unbroadcast(g, b)))


def sub(a, b):      # core/ops.py:61-62 — literally a + (-b): two nodes
    return add(a, neg(b))


def mul(a, b):      # core/ops.py:65-90
    return _node(a.values * b.values,
                 (a, lambda g: _unbroadcast(g * b.values, a)),
                 (b, lambda g: _unbroadcast(g * a.values, b)))


def div(a, b):      # core/ops.py:93-118
    return _node(a.values / b.values,
                 (a, lambda g: _unbroadcast(g / b.values, a)),
                 (b, lambda g: _unbroadcast(-g * a.values / b.values ** 2, b)))


def power(a, b):    # core/ops.py:121-147
    out = a.values ** b.values
    return _node(out,
                 (a, lambda g: _unbroadcast(g * b.values * a.values ** (b.values - 1), a)),
                 (b, lambda g: _unbroadcast(g * (np.log(a.values) * out), b)))


def dot(a, b):      # core/ops.py:150-163
    return _node(a.values @ b.values,
                 (a, lambda g: g @ b.values.T),
                 (b, lambda g: a.values.T @ g))


def maximum(a, b):  # core/ops.py:166-188 (ties -> a)
    a, b = _t(a), _t(b)
    return _node(np.maximum(a.values, b.values),
                 (a, lambda g: _unbroadcast(g * (a.values >= b.values), a)),
                 (b, lambda g: _unbroadcast(g * (b.values > a.values), b)))


def minimum(a, b):  # core/ops.py:191-213 (ties -> a)
    a, b = _t(a), _t(b)
    return _node(np.minimum(a.values, b.values),
                 (a, lambda g: _unbroadcast(g * (a.values <= b.values), a)),
                 (b, lambda g: _unbroadcast(g * (b.values < a.values), b)))


def exp(a):         # core/ops.py:216-222
    a = _t(a)
    out = np.exp(a.values)
    return _node(out, (a, lambda g: out * g))


def log(a):         # core/ops.py:243-249
    a = _t(a)
    return _node(np.log(a.values), (a, lambda g: g / a.values))


def neg(a):         # core/ops.py:293-299
    return _node(-a.values, (a, lambda g: -g))


def rmax(a, axis=None):   # core/ops.py:225-231 (all ties get the gradient)
    a = _t(a)
    return _node(np.max(a.values, axis=axis),
                 (a, lambda g: g * (a.values.max(axis=axis, keepdims=1) == a.values)))


def rmin(a, axis=None):   # core/ops.py:234-240
    a = _t(a)
    return _node(np.min(a.values, axis=axis),
                 (a, lambda g: g * (a.values.min(axis=axis, keepdims=1) == a.values)))


def rsum(a, axis=None):   # core/ops.py:252-265
    a = _t(a)

    def back(g):
        if axis is None:
            return g * np.ones_like(a.values)
        return np.repeat(np.expand_dims(g, axis), a.values.shape[axis], axis)

    return _node(a.values.sum(axis=axis), (a, back))


def transpose(a, axes=None):   # core/ops.py:268-279
    perm = list(reversed(range(a.values.ndim))) if axes is None else list(axes)
    return _node(a.values.transpose(axes), (a, lambda g: g.transpose(np.argsort(perm))))


def getitem(a, key):      # core/ops.py:282-290
    def back(g):
        full = np.zeros_like(a.values)
        full[key] = g
        return full
    return _node(a.values[key], (a, back))


def reshape(a, newshape):  # core/ops.py:302-309
    a = _t(a)
    shape = a.values.shape
    return _node(a.values.reshape(newshape), (a, lambda g: g.reshape(shape)))


def pad(a, pad_width, mode="constant"):   # core/ops.py:312-321
    a = _t(a)
    out = np.pad(a.values, pad_width=pad_width, mode=mode)
    window = tuple(slice(b, n - e) for n, (b, e) in zip(out.shape, pad_width))
    return _node(out, (a, lambda g: g[window]))


def flatten(a):           # core/ops.py:324-330
    a = _t(a)
    shape = a.shape
    return _node(a.values.ravel(), (a, lambda g: g.reshape(shape)))


def clip(a, lo=None, hi=None):   # core/ops.py:333-344 (mask built eagerly, inclusive bounds)
    a = _t(a)
    mask = np.ones(a.shape, dtype=bool)
    if lo is not None:
        mask &= a.values >= lo
    if hi is not None:
        mask &= a.values <= hi
    return _node(a.values.clip(lo, hi), (a, lambda g: g * mask))
