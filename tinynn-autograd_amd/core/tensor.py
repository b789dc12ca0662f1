"""Device-backed Tensor with the reference's autograd surface (reference: core/tensor.py:13-171).

Same constructor, attributes (`values`, `grad`, `requires_grad`, `dependency`, `shape`), operators
(including the non-autograd in-place forms that replace `values` and drop `grad`, core/tensor.py:35-38,
66-68) and `backward` / `zero_grad` semantics, but `values` and `grad` are DeviceArrays in HBM.

Differences that are deliberate:
  * floating data defaults to float32 on device (the reference silently promotes everything to float64,
    SURVEY F4); `set_default_float(np.float64)` restores bit-for-bit behaviour for the known-answer tests.
  * `backward()` called on a root schedules every reachable node ONCE in topological order and hands each
    node the SUM of the gradients arriving on all of its edges, instead of recursing once per edge
    (core/tensor.py:157-168 re-traverses the MNIST MLP 4x per step).  All vjps are linear in the
    incoming gradient, so `.grad` of every node ends up identical; gradients still accumulate across calls
    until `zero_grad()` (core/tensor.py:163, test/test_autograd.py:175-179).
  * zero gradients are lazy: `zero_grad()` records "zero" and the first accumulation adopts the incoming
    buffer instead of doing memset + add.
"""

import weakref

import numpy as np

from .. import _lib
from .. import device_array as da
from ..device_array import DeviceArray
from . import ops

_CAPTURE_LEAVES = {}      # id -> weakref of the leaf tensors that received a gradient while a hipGraph capture was open


def take_capture_grads():
    """End of a hipGraph capture (graph.py): the host-side gradient state of every leaf tensor (parameter) the captured
    function accumulated into, as (weakref, grad array, shared flag).  A replay refreshes those BUFFERS but not the Python
    attributes — and an eager `model.step()` between two replays drops them (`param.values = ...`, core/tensor.py:35-38) — so
    the captured function re-installs this state after every replay: `loss-only graph; model.step()` in a loop then sees
    the gradients of THAT replay, like the eager loop body of examples/mnist/run.py:79-83."""
    out = []
    for ref in _CAPTURE_LEAVES.values():
        t = ref()
        if t is not None and t._grad is not None:
            out.append((ref, t._grad, t._grad_shared))
    _CAPTURE_LEAVES.clear()
    return out


def as_tensor(obj):
    """reference: core/tensor.py:7-10"""
    if not isinstance(obj, Tensor):
        obj = Tensor(obj)
    return obj


def _broadcasts_to(src, dst):
    """numpy's rule for `dst_array += src_array` on shapes alone (np.broadcast_shapes(src, dst) == dst) — without building numpy
    objects: once per step on the loss seed of loss.backward(), 3-4 us of the eager step"""
    n = len(dst) - len(src)
    if n < 0:
        return False
    for k, d in enumerate(src):
        if d != 1 and d != dst[n + k]:
            return False
    return True


class Tensor(object):

    def __init__(self, values, requires_grad=False, dependency=None, dtype=None):
        self._values = da.asarray(values, dtype)
        self._grad = None
        self._grad_zero = False      # lazily-zero gradient (no buffer yet)
        self._grad_shared = False    # _grad aliases a buffer that other nodes may also hold
        self._grad_home = None       # optional pinned view (flat gradient arena) to accumulate into
        self._values_home = None     # optional pinned view (flat parameter arena)
        self.requires_grad = requires_grad
        if self.requires_grad:
            self.zero_grad()
        self.dependency = dependency
        if self.dependency is None:
            self.dependency = []
        self._fused_vjp = None       # optional: all edges' contributions from one launch (see backward)

    # ------------------------------------------------------------------ values / grad
    @property
    def values(self):
        return self._values

    @values.setter
    def values(self, new_values):
        # reference: core/tensor.py:35-38 — replacing the array also forgets the gradient
        new_values = da.asarray(new_values)
        if self._values_home is not None and new_values.shape == self._values_home.shape:
            if new_values is not self._values_home:
                self._values_home[...] = new_values      # stay inside the parameter arena
            self._values = self._values_home
        else:
            self._values = new_values
            self._values_home = None
        self._grad = None
        self._grad_zero = False
        self._grad_shared = False

    @property
    def grad(self):
        if self._grad is None and self._grad_zero:
            if self._grad_home is not None:
                self._grad_home.fill(0.0)
                self._grad = self._grad_home
            else:
                self._grad = da.zeros(self.shape, self._float_dtype())
            self._grad_zero = False
            self._grad_shared = False
        elif self._grad_shared and self._grad is not None:
            # the buffer was adopted from the backward pass and may be aliased by another node: hand out
            # a private copy so that in-place edits of `.grad` behave like the reference's own ndarray
            self._grad = self._grad.copy()
            self._grad_shared = False
        return self._grad

    @grad.setter
    def grad(self, value):
        self._grad = None if value is None else da.asarray(value)
        self._grad_zero = False
        self._grad_shared = value is not None

    def _float_dtype(self):
        return self._values.dtype if self._values.dtype.kind == "f" else da.get_default_float()

    @property
    def shape(self):
        return self._values.shape

    def __repr__(self):
        return "Tensor(shape=%s, requires_grad=%s)" % (self.shape, self.requires_grad)

    def __len__(self):
        return len(self._values)

    # ------------------------------------------------------------------ comparisons: raw bool arrays
    def __gt__(self, other):
        return self.values > as_tensor(other).values

    def __lt__(self, other):
        return self.values < as_tensor(other).values

    def __ge__(self, other):
        return self.values >= as_tensor(other).values

    def __le__(self, other):
        return self.values <= as_tensor(other).values

    # ------------------------------------------------------------------ differentiable operators
    def __add__(self, other):
        return ops.add_(self, as_tensor(other))

    def __radd__(self, other):
        return ops.add_(as_tensor(other), self)

    def __sub__(self, other):
        return ops.sub_(self, as_tensor(other))

    def __rsub__(self, other):
        return ops.sub_(as_tensor(other), self)

    def __mul__(self, other):
        return ops.mul_(self, as_tensor(other))

    def __rmul__(self, other):
        return ops.mul_(as_tensor(other), self)

    def __truediv__(self, other):
        return ops.div_(self, as_tensor(other))

    def __rtruediv__(self, other):
        return ops.div_(as_tensor(other), self)

    def __pow__(self, other):
        return ops.pow_(self, as_tensor(other))

    def __rpow__(self, other):
        return ops.pow_(as_tensor(other), self)

    def __matmul__(self, other):
        return ops.dot_(self, as_tensor(other))

    def __rmatmul__(self, other):
        return ops.dot_(as_tensor(other), self)

    def __neg__(self):
        return ops.neg_(self)

    def __getitem__(self, key):
        return ops.getitem_(self, key)

    # ------------------------------------------------------------------ non-autograd in-place forms
    def _inplace(self, fn, other):
        other = as_tensor(other).values
        if self._values_home is not None and self._values is self._values_home:
            fn(self._values, other, True)          # update the arena slice where it lives
            self.values = self._values
        else:
            self.values = fn(self._values, other, False)
        return self

    def __iadd__(self, other):
        return self._inplace(lambda a, b, ip: a.__iadd__(b) if ip else a + b, other)

    def __isub__(self, other):
        return self._inplace(lambda a, b, ip: a.__isub__(b) if ip else a - b, other)

    def __imul__(self, other):
        return self._inplace(lambda a, b, ip: a.__imul__(b) if ip else a * b, other)

    def __itruediv__(self, other):
        return self._inplace(lambda a, b, ip: a.__itruediv__(b) if ip else a / b, other)

    def __ipow__(self, other):
        self.values = self.values ** as_tensor(other).values
        return self

    def __imatmul__(self, other):
        self.values = self.values @ as_tensor(other).values
        return self

    # ------------------------------------------------------------------ method forms of ops
    def sum(self, axis=None):
        return ops.sum_(self, axis=axis)

    def max(self, axis=None):
        return ops.max_(self, axis=axis)

    def min(self, axis=None):
        return ops.min_(self, axis=axis)

    def transpose(self, axes=None):
        return ops.transpose_(self, axes=axes)

    def log(self):
        return ops.log_(self)

    def exp(self):
        return ops.exp_(self)

    def reshape(self, newshape):
        return ops.reshape_(self, newshape)

    def flatten(self):
        return ops.flatten_(self)

    def clip(self, min=None, max=None):
        return ops.clip_(self, min, max)

    @property
    def T(self):
        return ops.transpose_(self, axes=None)

    # ------------------------------------------------------------------ numpy protocols
    # np.exp(-x) on a Tensor raises in the reference (core/layers.py:79-80); routing ufuncs
    # to the differentiable ops makes the unmodified Sigmoid layer work.
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            return NotImplemented
        table = _UFUNC_TO_OP
        if ufunc in table:
            return table[ufunc](*[as_tensor(x) for x in inputs])
        return NotImplemented

    def __array_function__(self, func, types, args, kwargs):
        if func is np.argmax:
            # examples/mnist/run.py:89 — integer class ids go back to the host like the reference's
            a = args[0].values if isinstance(args[0], Tensor) else args[0]
            return np.asarray(da.argmax(a, *args[1:], **kwargs))
        if func is np.shape:
            return self.shape
        return NotImplemented

    def __array__(self, dtype=None, copy=None):
        return self._values.__array__(dtype)

    # ------------------------------------------------------------------ autograd
    def zero_grad(self):
        """reference: core/tensor.py:170-171 (np.zeros(self.shape)) — recorded lazily."""
        self._grad = None
        self._grad_zero = True
        self._grad_shared = False
        self._home_lent = False
        home = self._grad_home
        if home is not None and type(home) is da.LazyArray:
            home.drop()                              # a deferred backward launch (core/model.py) is not wanted any more

    def _accumulate(self, g):
        """self.grad += g (core/tensor.py:163) with broadcasting of g to self.shape."""
        if self._grad is None and not self._grad_zero:
            raise TypeError("unsupported operand type(s) for +=: 'NoneType' and 'DeviceArray' "
                            "(the gradient was dropped by a value assignment; call zero_grad())")
        if not isinstance(g, da.DeviceArray):
            g = da.asarray(g)
        self._home_lent = False                      # see ops.dense_: the arena view may be lent to a fused vjp
        if g.shape != self.shape or g.dtype != self._float_dtype() or g.is_host_scalar:
            if not _broadcasts_to(g.shape, self.shape):
                raise ValueError("non-broadcastable output operand with shape %s doesn't match the "
                                 "broadcast shape" % (self.shape,))
            g = g.astype(self._float_dtype())._broadcast_to(self.shape)
        if self._grad is None:                       # lazily zero: adopt instead of memset + add
            if self._grad_home is not None:
                if g is not self._grad_home:         # a vjp with `into` already wrote the arena view itself
                    self._grad_home[...] = g
                self._grad, self._grad_shared = self._grad_home, False
            else:
                self._grad, self._grad_shared = g, True
            self._grad_zero = False
        elif self._grad_shared:
            self._grad, self._grad_shared = self._grad + g, False
        else:
            self._grad += g
        if _lib.capturing and not self.dependency:
            _CAPTURE_LEAVES[id(self)] = weakref.ref(self)

    def backward(self, grad=None):
        assert self.requires_grad, "Call backward() on a non-requires-grad tensor."
        seed = da.asarray(1.0 if grad is None else grad)

        # iterative post-order DFS over the dependency edges -> reverse = topological order from the root
        order, seen, stack = [], {id(self)}, [(self, 0)]
        while stack:
            node, i = stack.pop()
            if i < len(node.dependency):
                stack.append((node, i + 1))
                child = node.dependency[i]["tensor"]
                if id(child) not in seen:
                    seen.add(id(child))
                    stack.append((child, 0))
            else:
                order.append(node)

        pending = {id(self): seed}

        def lendable(child):
            """The child's arena view, if this is the first contribution of a lazily-zero, arena-backed leaf."""
            home = getattr(child, "_grad_home", None)
            if (home is not None and id(child) not in pending and not child.dependency
                    and child._grad is None and child._grad_zero):
                return home
            return None

        for node in reversed(order):
            g = pending.pop(id(node), None)
            if g is None:
                continue
            node._accumulate(g)
            if not node.dependency:
                continue

            # a node may offer ALL its edges from one launch (ops.dense_: dW + db + masked dX)
            contribs, fused = None, getattr(node, "_fused_vjp", None)
            if fused is not None and len({id(d["tensor"]) for d in node.dependency}) == len(node.dependency):
                contribs = fused(g, [lendable(d["tensor"]) for d in node.dependency])
            for j, dep in enumerate(node.dependency):
                child, fn = dep["tensor"], dep["grad_fn"]
                key = id(child)
                if contribs is not None:
                    contrib = contribs[j]
                else:
                    # a vjp may offer a form that writes straight into the leaf's arena view (`fn.into`; for a bound
                    # method the attribute sits on the function, so it is bound to the same object here)
                    into = getattr(fn, "into", None)
                    if into is not None and lendable(child) is not None:
                        owner = getattr(fn, "__self__", None)
                        contrib = (into(owner, g, child._grad_home) if owner is not None
                                   else into(g, child._grad_home))
                    else:
                        contrib = fn(g)
                if key in pending:
                    pending[key] = pending[key] + contrib
                else:
                    pending[key] = contrib


_UFUNC_TO_OP = {
    np.exp: lambda a: ops.exp_(a),
    np.log: lambda a: ops.log_(a),
    np.negative: lambda a: ops.neg_(a),
    np.add: lambda a, b: ops.add_(a, b),
    np.subtract: lambda a, b: ops.sub_(a, b),
    np.multiply: lambda a, b: ops.mul_(a, b),
    np.true_divide: lambda a, b: ops.div_(a, b),
    np.power: lambda a, b: ops.pow_(a, b),
    np.maximum: lambda a, b: ops.maximum_(a, b),
    np.minimum: lambda a, b: ops.minimum_(a, b),
    np.matmul: lambda a, b: ops.dot_(a, b),
}
