"""Differentiable ops on device-backed Tensors (reference: core/ops.py:12-384).

Every public name of the reference module exists here with the same signature and the same vjp
semantics (tie rules of maximum_/minimum_/max_/min_, inclusive clip mask, un-broadcast rule, saved
inputs held by reference in the closures).  Forward values and vjps are DeviceArray expressions, i.e.
HIP kernels behind the C-ABI; nothing is computed on the host.  Cited line numbers point at the
reference expression each piece replaces.

Extra fused nodes used by this package's own layers/losses (parity-tested against the generic chain):
`dense_` (GEMM + bias epilogue), `sigmoid_`, `softmax_nll_` (whole-batch softmax NLL).
"""

import math
import weakref

import numpy as np

from .. import _lib
from .. import device_array as da


_AS_TENSOR = None


def as_tensor(obj):
    global _AS_TENSOR
    if _AS_TENSOR is None:
        from .tensor import as_tensor as _as_tensor   # lazy: tensor imports ops (resolved once, not on every call)
        _AS_TENSOR = _as_tensor
    return _AS_TENSOR(obj)


# ---------------------------------------------------------------------- graph construction
def build_binary_ops_tensor(ts1, ts2, grad_fn_ts1, grad_fn_ts2, values):
    """reference: core/ops.py:12-20"""
    parents = [(ts1, grad_fn_ts1), (ts2, grad_fn_ts2)]
    return _make_node(ts1.__class__, values, parents)


def build_unary_ops_tensor(ts, grad_fn, values):
    """reference: core/ops.py:23-29"""
    return _make_node(ts.__class__, values, [(ts, grad_fn)])


def _make_node(cls, values, parents):
    # only inputs that require grad become edges, so e.g. dX of the first Dense layer is never computed
    edges = [dict(tensor=t, grad_fn=fn) for t, fn in parents if t.requires_grad]
    return cls(values, bool(edges), edges)


def _unbroadcast(grad, shape):
    """Undo numpy broadcasting of an operand of `shape` (core/ops.py:41-46): sum away the leading axes
    the operand did not have, then sum (keepdims) over the axes where it had extent 1."""
    grad = da.asarray(grad)
    if grad.shape == tuple(shape):
        return grad
    extra = grad.ndim - len(shape)
    if extra > 0:
        grad = grad.sum(axis=tuple(range(extra)))
    for axis, dim in enumerate(shape):
        if dim == 1 and grad.shape[axis] != 1:
            grad = grad.sum(axis=axis, keepdims=True)
    return grad


# ---------------------------------------------------------------------- arithmetic
def add_(ts1, ts2):
    """reference: core/ops.py:32-58"""
    values = ts1.values + ts2.values
    return build_binary_ops_tensor(
        ts1, ts2,
        lambda g: _unbroadcast(g, ts1.shape),
        lambda g: _unbroadcast(g, ts2.shape),
        values)


def sub_(ts1, ts2):
    """reference: core/ops.py:61-62 (a + (-b)); one node here, same values and vjps."""
    values = ts1.values - ts2.values
    return build_binary_ops_tensor(
        ts1, ts2,
        lambda g: _unbroadcast(g, ts1.shape),
        lambda g: _unbroadcast(-da.asarray(g), ts2.shape),
        values)


def mul_(ts1, ts2):
    """reference: core/ops.py:65-90"""
    values = ts1.values * ts2.values
    return build_binary_ops_tensor(
        ts1, ts2,
        lambda g: _unbroadcast(g * ts2.values, ts1.shape),
        lambda g: _unbroadcast(g * ts1.values, ts2.shape),
        values)


def div_(ts1, ts2):
    """reference: core/ops.py:93-118"""
    values = ts1.values / ts2.values
    return build_binary_ops_tensor(
        ts1, ts2,
        lambda g: _unbroadcast(g / ts2.values, ts1.shape),
        lambda g: _unbroadcast(-da.asarray(g) * ts1.values / ts2.values ** 2, ts2.shape),
        values)


def pow_(ts1, ts2):
    """reference: core/ops.py:121-147"""
    values = ts1.values ** ts2.values

    def grad_base(g):
        return _unbroadcast(g * ts2.values * ts1.values ** (ts2.values - 1), ts1.shape)

    def grad_exponent(g):
        return _unbroadcast(g * (da.log(ts1.values) * values), ts2.shape)

    return build_binary_ops_tensor(ts1, ts2, grad_base, grad_exponent, values)


def dot_(ts1, ts2):
    """reference: core/ops.py:150-163.  `.T` is a lazy flag, so the vjps run as NT / TN GEMMs."""
    values = ts1.values @ ts2.values
    return build_binary_ops_tensor(
        ts1, ts2,
        lambda g: da.asarray(g) @ ts2.values.T,
        lambda g: ts1.values.T @ da.asarray(g),
        values)


def maximum_(ts1, ts2):
    """reference: core/ops.py:166-188 — ties send the gradient to ts1 (>=)."""
    values = da.maximum(ts1.values, ts2.values)
    return build_binary_ops_tensor(
        ts1, ts2,
        lambda g: _unbroadcast(da.mul_mask(g, ts1.values >= ts2.values), ts1.shape),
        lambda g: _unbroadcast(da.mul_mask(g, ts2.values > ts1.values), ts2.shape),
        values)


def minimum_(ts1, ts2):
    """reference: core/ops.py:191-213 — ties send the gradient to ts1 (<=)."""
    values = da.minimum(ts1.values, ts2.values)
    return build_binary_ops_tensor(
        ts1, ts2,
        lambda g: _unbroadcast(da.mul_mask(g, ts1.values <= ts2.values), ts1.shape),
        lambda g: _unbroadcast(da.mul_mask(g, ts2.values < ts1.values), ts2.shape),
        values)


# ---------------------------------------------------------------------- unary
def exp_(ts):
    """reference: core/ops.py:216-222 (the vjp reuses the saved output)"""
    values = da.exp(ts.values)
    return build_unary_ops_tensor(ts, lambda g: values * g, values)


def log_(ts):
    """reference: core/ops.py:243-249"""
    values = da.log(ts.values)
    return build_unary_ops_tensor(ts, lambda g: g / ts.values, values)


def neg_(ts):
    """reference: core/ops.py:293-299"""
    return build_unary_ops_tensor(ts, lambda g: -da.asarray(g), -ts.values)


def _extreme(ts, axis, rmax):
    x = ts.values
    values = x.max(axis=axis) if rmax else x.min(axis=axis)

    def grad_fn(g):
        # every element equal to the extreme receives the full gradient (core/ops.py:229,238)
        ext = x.max(axis=axis, keepdims=True) if rmax else x.min(axis=axis, keepdims=True)
        return da.mul_mask(g, ext == x)

    return build_unary_ops_tensor(ts, grad_fn, values)


def max_(ts, axis=None):
    """reference: core/ops.py:225-231"""
    return _extreme(ts, axis, True)


def min_(ts, axis=None):
    """reference: core/ops.py:234-240"""
    return _extreme(ts, axis, False)


def sum_(ts, axis):
    """reference: core/ops.py:252-265"""
    x = ts.values
    values = x.sum(axis=axis)

    def grad_fn(g):
        g = da.asarray(g)
        if axis is None:
            if g.ndim > x.ndim:
                raise ValueError("gradient of a full sum must be broadcastable to the input")
            return g._as_float(x.dtype if x.dtype.kind == "f" else None)._broadcast_to(x.shape)
        return np.expand_dims(g._contig(), axis)._broadcast_to(x.shape)

    return build_unary_ops_tensor(ts, grad_fn, values)


def transpose_(ts, axes=None):
    """reference: core/ops.py:268-279"""
    values = ts.values.transpose(axes)
    perm = list(reversed(range(ts.values.ndim))) if axes is None else [int(a) for a in axes]
    inverse = [int(i) for i in np.argsort(perm)]
    return build_unary_ops_tensor(ts, lambda g: da.asarray(g).transpose(inverse), values)


def getitem_(ts, key):
    """reference: core/ops.py:282-290 (slice = view, integer array = row gather kernel)"""
    x = ts.values
    if isinstance(key, ts.__class__):
        key = key.values
    values = x[key]

    def grad_fn(g):
        recovered = da.zeros(x.shape, x.dtype if x.dtype.kind == "f" else None)
        recovered[key] = g
        return recovered

    return build_unary_ops_tensor(ts, grad_fn, values)


def reshape_(ts, newshape):
    """reference: core/ops.py:302-309"""
    shape = ts.values.shape
    return build_unary_ops_tensor(ts, lambda g: da.asarray(g).reshape(shape), ts.values.reshape(newshape))


def pad_(ts, pad_width, mode):
    """reference: core/ops.py:312-321"""
    values = np.pad(ts.values, pad_width=pad_width, mode=mode)
    window = tuple(slice(int(before), int(size - after))
                   for size, (before, after) in zip(values.shape, pad_width))
    return build_unary_ops_tensor(ts, lambda g: da.asarray(g)[window], values)


def flatten_(ts):
    """reference: core/ops.py:324-330"""
    shape = ts.shape
    return build_unary_ops_tensor(ts, lambda g: da.asarray(g).reshape(shape), ts.values.ravel())


def clip_(ts, min, max):
    """reference: core/ops.py:333-344.  The inclusive mask (x >= min) & (x <= max) is recomputed from the
    saved input inside the vjp kernel instead of being stored as a bool array."""
    x = ts.values
    values = da.clip(x, min, max)
    return build_unary_ops_tensor(ts, lambda g: da.clip_bwd(g, x, min, max), values)


# ---------------------------------------------------------------------- fused nodes (this package's own)
class _DenseVjp(object):
    """The vjps of one ops.dense_ node (see dense_): per-edge forms for schedulers that ask edge by edge (the reference's
    own recursion), `fused_vjp` for Tensor.backward, which offers all edges at once."""
    __slots__ = ("b", "xv", "wv", "bv", "out", "m", "k", "n", "dt", "relu", "x_relu", "edges", "shared")

    def __init__(self, b, xv, wv, bv, out, m, k, n, dt, relu, x_relu):
        # arrays, not the input tensors (only the bias tensor, a leaf, for its arena view): a vjp that held its node's input
        # TENSOR closed a reference cycle through the speculative head (x -> its wrapped vjp -> the classifier's vjp -> x)
        self.b, self.xv, self.wv, self.bv, self.out = b, xv, wv, bv, out
        self.m, self.k, self.n, self.dt, self.relu, self.x_relu = m, k, n, dt, relu, x_relu
        self.edges = ()
        self.shared = []              # [(g object, db)] computed together with dW, waiting for the bias vjp

    def dz_of(self, g_in):
        """Gradient w.r.t. the pre-activation: g itself, or g * [z >= 0] for the fused ReLU unless a consumer's dX
        launch applied that mask already."""
        g = da.asarray(g_in)
        if not self.relu or g._tag is self.out:
            return g
        return da.mul_signmask(g, self.out)

    def d_x(self, g):
        return self.dz_of(g) @ self.wv.T

    def _dw_db(self, g_in, dest):
        m, k, n, dt, b = self.m, self.k, self.n, self.dt, self.b
        g = self.dz_of(g_in)
        if g.shape != (m, n) or g.dtype != dt or g._hv is not None or not g.size:
            return None
        g = g._contig()
        dw = dest if dest is not None else da.empty((k, n), dt)
        db = None
        if b.requires_grad:
            home = getattr(b, "_grad_home", None)
            if (home is not None and b._grad is None and b._grad_zero and not getattr(b, "_home_lent", False)
                    and home.size == n and home.dtype == dt):
                db = home                            # the bias' own arena view: lent out once per backward
                b._home_lent = True
            else:
                db = da.empty(tuple(b.shape), dt)
        _lib.get().gemm_tn_colsum(k, n, m, self.xv._ptr, k, g._ptr, n, dw._ptr, n, db._ptr if db is not None else None,
                                  dw._code())
        if db is not None:
            del self.shared[:]
            self.shared.append((g_in, db))
        return dw

    def d_w(self, g):
        dw = self._dw_db(g, None)
        return dw if dw is not None else self.xv.T @ self.dz_of(g)

    def d_w_into(self, g, dest):
        if dest.shape != (self.k, self.n) or dest.dtype != self.dt or dest._t or dest._hv is not None:
            return self.d_w(g)
        dw = self._dw_db(g, dest)
        return dw if dw is not None else self.d_w(g)

    d_w.into = d_w_into               # Tensor.backward: the first contribution of a lazily-zero leaf goes to its arena view

    def d_b(self, g):
        if self.shared and self.shared[0][0] is g:
            return self.shared.pop()[1]
        return _unbroadcast(self.dz_of(g), self.b.shape)

    def fused_vjp(self, g_in, homes):
        """All edges from ONE tnn_dense_bwd launch.  homes[i]: the arena view of edge i's tensor when the scheduler
        lends it (first contribution of a lazily-zero leaf), else None.  Returns None to decline (odd gradients)."""
        m, k, n, dt, edges = self.m, self.k, self.n, self.dt, self.edges
        g = da.asarray(g_in)
        if g.shape != (m, n) or g.dtype != dt or g._hv is not None or g._t or not g.size:
            return None
        if "x" in edges and not self.x_relu:
            return None                               # dX without a mask epilogue: the per-edge GEMMs
        g = self.dz_of(g)._contig()
        xv = self.xv
        res = {}
        dw = db = dx = None
        for name, home in zip(edges, homes):
            if name == "w":
                ok = home is not None and home.shape == (k, n) and home.dtype == dt and not home._t and home._hv is None
                dw = home if ok else da.empty((k, n), dt)
                res["w"] = dw
            elif name == "b":
                ok = home is not None and home.size == n and home.dtype == dt and not home._t and home._hv is None
                db = home if ok else da.empty(tuple(self.b.shape), dt)
                res["b"] = db
            else:
                dx = da.empty((m, k), dt)
                dx._tag = xv                          # already multiplied by x's ReLU mask
                res["x"] = dx
        if dw is None:
            dw = da.empty((k, n), dt)                 # w frozen: the launch still needs somewhere to write
        if dx is None and edges == ["w", "b"] and type(dw) is da.LazyArray and type(db) is da.LazyArray:
            # the model's FIRST layer writing into its arena views: the launch can wait for the optimizer (core/model.py)
            owner = getattr(self.b, "_defer_first", None)
            owner = owner() if owner is not None else None
            if owner is not None and owner._defer_first_backward(xv, g, self.wv, dw, db, m, k, n):
                return [dw, db]
        _lib.get().dense_bwd(m, k, n, xv._ptr, g._ptr, self.wv._ptr, dw._ptr, db._ptr if db is not None else None,
                             dx._ptr if dx is not None else None, xv._ptr if dx is not None else None, dw._code())
        return [res[name] for name in edges]


def dense_(x, w, b, relu=False, head_w=None, lazy=False, head_b=None):
    """x @ w + b as ONE GEMM with a bias epilogue (core/layers.py:49); the vjps are the NT / TN GEMMs of
    dot_ (core/ops.py:156-160) and the column-sum of add_'s un-broadcast (:49-55).

    relu=True (Net.forward fuses a Dense that is directly followed by a ReLU layer): the epilogue also applies
    clip(., 0) (core/layers.py:97-98) and keeps the vjp mask z >= 0 (core/ops.py:338, gradient 1 AT zero) in the sign
    bit of zero — z < 0 is stored as -0.0, z >= 0 as |z| — so no pre-activation and no mask array exist.

    Backward is ONE launch per layer whenever the scheduler (Tensor.backward) offers all edges at once
    (`_fused_vjp`): tnn_dense_bwd writes dW and db (straight into the parameters' arena views when they are lent)
    and, if x is itself a sign-encoded ReLU output, dX = (g W^T) * [z_prev >= 0] from the same launch.  That dX is
    tagged as already masked for x's producer, whose own vjp then skips its mask pass; a gradient that reaches a ReLU
    node untagged (several consumers were summed, or a foreign consumer) is masked there — masking twice is harmless
    (the mask is idempotent), so the rule never changes a value.

    head_w (Net.forward, hidden layer in front of a classifier head): the launch also emits the NEXT layer's logits as
    per-tile partial sums (tnn_dense_fwd_head_partials) — they ride on the output array (`_aux`) for the loss node.  More than
    128 rows (head_b, the classifier's bias, is then needed too): the row-panel launch instead (tnn_dense_fwd_rows_head_stats:
    whole logits without the bias + one {max, sum-exp} pair per 16 rows).
    lazy (Net.forward, the classifier layer itself in TRAIN mode): the GEMM is deferred until the logits are first used
    (device_array.LazyArray); softmax_nll_ then produces them together with the loss and this layer's backward."""
    xv, wv, bv = x.values, w.values, b.values
    if xv.ndim != 2 or wv.ndim != 2 or xv.shape[1] != wv.shape[0] or bv.size != wv.shape[1]:
        raise ValueError("dense_: shapes %s @ %s + %s do not line up" % (xv.shape, wv.shape, bv.shape))
    m, k = xv.shape
    n = wv.shape[1]
    dt = da._float_result_dtype(xv, wv)
    x_relu = xv._tag is da.RELU_SIGN and xv.dtype == dt and not xv._t and xv._hv is None
    xc = xv._as_float(dt)._contig()
    x_relu = x_relu and xc is xv                      # the sign bits must be the producer's own buffer
    xv, wv, bv = xc, wv._as_float(dt)._contig(), bv._as_float(dt)._contig()
    code = da._CODE[dt]
    if lazy and not relu and m and n and dt == np.float32:
        def produce(arr, xv=xv, wv=wv, bv=bv):
            _lib.get().gemm_bias_act(0, 0, m, n, k, xv._ptr, k, wv._ptr, n, bv._ptr, _lib.ACT_NONE, 0,
                                     da.DeviceArray._ptr.__get__(arr, da.DeviceArray), n, code)
        out = da.LazyArray.deferred((m, n), dt, produce)
    else:
        lazy = False
        out = da.empty((m, n), dt)
        hw = head_w.values if head_w is not None else None
        hb = head_b.values if head_b is not None else None
        if (hw is not None and 128 < m <= 1024 and relu and n == 128 and k % 4 == 0 and dt == np.float32 and hw.dtype == dt
                and hw.shape == (128, 10) and not hw._t and hw._hv is None and hb is not None and hb.dtype == dt
                and hb.size == 10 and not hb._t and hb._hv is None):
            zfull, pairs = da.empty((m, 10), dt), da.empty(((m + 15) // 16, 2), dt)
            _lib.get().dense_fwd_rows_head_stats(m, n, k, xv._ptr, k, wv._ptr, n, bv._ptr, _lib.ACT_RELU, 1, out._ptr, n,
                                                 hw._ptr, 10, zfull._ptr, hb._ptr, pairs._ptr, code)
            out._aux = (zfull, hw, pairs, hb)         # valid for the classifier weights / bias as they are NOW
        elif (hw is not None and out.size and m <= 128 and dt == np.float32 and hw.dtype == dt and hw.ndim == 2 and hw.shape[0] == n
                and hw.shape[1] <= 16 and not hw._t and hw._hv is None):
            zpart = da.empty(((n + 15) // 16, m, hw.shape[1]), dt)
            _lib.get().dense_fwd_head_partials(m, n, k, xv._ptr, k, wv._ptr, n, bv._ptr, _lib.ACT_RELU if relu else _lib.ACT_NONE,
                                               1 if relu else 0, out._ptr, n, hw._ptr, hw.shape[1], zpart._ptr, code)
            out._aux = (zpart, hw)                    # valid for the classifier weights `hw` as they are NOW
        elif out.size:
            _lib.get().gemm_bias_act(0, 0, m, n, k, xv._ptr, k, wv._ptr, n, bv._ptr,
                                     _lib.ACT_RELU if relu else _lib.ACT_NONE, 1 if relu else 0, out._ptr, n, code)
    if relu:
        out._tag = da.RELU_SIGN

    # the vjps live on ONE context object (bound methods) instead of eight closures per call: the op-level step is host-bound
    ctx = _DenseVjp(b, xv, wv, bv, out, m, k, n, dt, relu, x_relu)
    node = _make_node(x.__class__, out, [(x, ctx.d_x), (w, ctx.d_w), (b, ctx.d_b)])
    edges = ctx.edges = [name for name, t in (("x", x), ("w", w), ("b", b)) if t.requires_grad]
    fused_vjp = ctx.fused_vjp
    node._fused_vjp = fused_vjp
    if lazy:
        node._head = (x, w, b, xv, wv, bv, edges)     # what softmax_nll_ needs to run the head in one launch
    elif relu:
        node._dense_ctx = (x, w, b, xv, wv, bv, edges, x_relu)   # ... and the hidden layer's backward with it
    return node


def _softmax_head(logits, labels):
    """The classifier head as ONE launch when the logits are still pending (dense_(lazy=True)) and the shapes are the ones
    tnn_mlp_head_tick takes: last Dense forward (core/layers.py:49) + whole-batch softmax NLL (core/losses.py:24-32) + the last
    Dense's backward (core/ops.py:156-160, :52-54, ReLU mask :342-343), i.e. three of the op-level step's launches.  The
    backward part is SPECULATIVE: its results are handed over when backward() reaches the Dense node with the loss node's own
    dz (the default seed 1.0); any other seed, a second backward or a foreign consumer recomputes through the ordinary vjps.
    Returns None when the fusion does not apply."""
    head = getattr(logits, "_head", None)
    z = logits._values
    if head is None or type(z) is not da.LazyArray or not z.pending:
        return None
    x, w, b, xv, wv, bv, edges = head
    y = as_tensor(labels).values
    m, c = z.shape
    hdim = wv.shape[0]
    aux = xv._aux
    # more than 128 rows: only with what the row-panel forward left behind (whole logits + panel statistics for THESE weights)
    rows_form = (m > 128 and aux is not None and len(aux) == 4 and aux[1] is wv and aux[3] is bv
                 and aux[0].shape == (m, c) and hdim == 128 and c == 10)
    if (y.shape != z.shape or xv._tag is not da.RELU_SIGN or xv._t or xv._hv is not None or wv._t or bv._t
            or not (rows_form or (m <= 128 and _head_fits(m, hdim, c)))):
        return None
    dt = np.dtype(np.float32)
    y = y._as_float(dt)._contig()
    lib = _lib.get()
    stats, loss, dz = da.empty((2,), dt), da.empty((), dt), da.empty((m, c), dt)

    # Speculative results may be written straight into a parameter's arena view, but a LATER loss evaluation (second forward +
    # loss before this one's backward) would write ITS results into the same view: every view written here is stamped with this
    # call's token on its tensor, and the hand-over below adopts a view only while the stamp is still this call's.
    owner = object()

    def dest(t, shape):
        """The tensor's arena view when its gradient is lazily zero (what the scheduler would lend), else a fresh buffer."""
        home = getattr(t, "_grad_home", None)
        if (t.requires_grad and home is not None and t._grad is None and t._grad_zero and not t.dependency
                and home.size == math.prod(shape) and home.dtype == dt and not home._t and home._hv is None):
            t._spec_owner = owner
            return home, True
        return da.empty(shape, dt), False
    dw, dw_home = dest(w, (hdim, c))
    db, db_home = dest(b, tuple(b.shape))
    zpart = None
    if rows_form:
        zpart = aux[0]
    elif aux is not None and aux[1] is wv and aux[0].shape == (hdim // 16, m, c):
        zpart = aux[0]
    generic = logits._fused_vjp

    # ---- with the hidden layer's backward in the same launch (tnn_mlp_head_bwd_tick: the 4-launch trainer's third launch)
    # when that layer is a fused Dense+ReLU node whose own input carries a ReLU mask: its dz is derived inside the tiles and
    # never stored, so the gradient handed to the hidden activation is a DEFERRED array (computed by the ordinary vjp only
    # if somebody reads that intermediate .grad) which the hidden node recognises and answers with the precomputed results
    hid = getattr(x, "_dense_ctx", None)
    if (hid is not None and zpart is not None and x.requires_grad and len(x.dependency) == len(hid[6])
            and hid[7] and hid[3].shape[1] % 16 == 0 and hid[4].dtype == dt and not hid[4]._t and "x" in hid[6]):
        x0, w1, b1, x0v, w1v, b1v, edges1, _ = hid
        n_in = x0v.shape[1]
        dw1, dw1_home = dest(w1, (n_in, hdim))
        db1, db1_home = dest(b1, tuple(b1.shape))
        dx0 = da.empty((m, n_in), dt)
        dx0._tag = x0v
        # Adam's bias-correction powers for the coming step are advanced by one thread of this launch when the classifier's
        # parameters belong to a Model whose optimizer asks for it (once per optimizer step: Adam.take_tick)
        opt = getattr(w, "_tick_optimizer", None)
        opt = opt() if opt is not None else None
        tick = opt.take_tick() if opt is not None else None
        pows, tb1, tb2 = tick if tick is not None else (None, 0.0, 0.0)
        if rows_form:
            pairs = aux[2]
            lib.mlp_head_bwd_tick_ext(m, m, n_in, hdim, c, x0v._ptr, w1v._ptr, xv._ptr, wv._ptr, bv._ptr, y._ptr, zpart._ptr,
                                      pairs._ptr, -pairs.shape[0], z.fulfilled_ptr(), dz._ptr, stats._ptr, loss._ptr, dw._ptr,
                                      db._ptr, dw1._ptr, db1._ptr, dx0._ptr, _lib.F32, pows, tb1, tb2)
        else:
            lib.mlp_head_bwd_tick(m, n_in, hdim, c, x0v._ptr, w1v._ptr, xv._ptr, wv._ptr, bv._ptr, y._ptr, zpart._ptr,
                                  z.fulfilled_ptr(), dz._ptr, stats._ptr, loss._ptr, dw._ptr, db._ptr, dw1._ptr, db1._ptr,
                                  dx0._ptr, _lib.F32, pows, tb1, tb2)

        def hidden_dz(arr, dz=dz, wv=wv, xv=xv):     # only if the intermediate gradient is actually looked at
            val = da.mul_signmask(dz @ wv.T, xv)
            _lib.get().memcpy_d2d(da.DeviceArray._ptr.__get__(arr, da.DeviceArray), val._ptr, val.nbytes)
        dx = da.LazyArray.deferred((m, hdim), dt, hidden_dz)
        dx._tag = xv
        pre1 = {"x": (dx0, False), "w": (dw1, dw1_home), "b": (db1, db1_home)}
        stamped1 = {"w": w1, "b": b1}
        generic1 = x._fused_vjp

        x_ref = weakref.ref(x)                        # not x itself: x._fused_vjp -> this closure -> x would be a cycle that
                                                      # keeps the whole step's graph (and its buffers) alive until the cycle GC

        def hidden_vjp(g_in, homes):
            x = x_ref()
            state = x._head_pre
            x._head_pre = None
            if state is not None and g_in is dx:
                out = []
                for name, home in zip(edges1, homes):
                    arr, wrote_home = pre1[name]
                    if wrote_home and (home is not arr or getattr(stamped1[name], "_spec_owner", None) is not owner):
                        out = None                    # not this tensor's to adopt, or overwritten by a later loss evaluation
                        break
                    out.append(arr)
                if out is not None:
                    return out
            return generic1(g_in, homes)
        x._head_pre = True
        x._fused_vjp = hidden_vjp
    elif rows_form:
        return None                                   # the one-launch head without the hidden backward exists for <= 128 rows only
    else:
        dx = da.empty((m, hdim), dt) if x.requires_grad else None
        lib.mlp_head_tick(m, hdim, c, xv._ptr, wv._ptr, bv._ptr, y._ptr, None if zpart is None else zpart._ptr,
                          z.fulfilled_ptr(), dz._ptr, stats._ptr, loss._ptr, dw._ptr, db._ptr, None if dx is None else dx._ptr,
                          _lib.F32, None, 0.0, 0.0)
        if dx is not None:
            dx._tag = xv                              # already multiplied by x's ReLU mask
    pre = {"x": (dx, False), "w": (dw, dw_home), "b": (db, db_home)}
    stamped = {"w": w, "b": b}

    logits_ref = weakref.ref(logits)                  # see x_ref above

    def fused_vjp(g_in, homes):
        logits = logits_ref()
        state = logits._head_pre
        logits._head_pre = None                       # one hand-over; whatever comes later is recomputed
        if state is not None and g_in is dz:
            out = []
            for name, home in zip(edges, homes):
                arr, wrote_home = pre[name]
                if wrote_home and (home is not arr or getattr(stamped[name], "_spec_owner", None) is not owner):
                    out = None                        # the arena view is no longer this tensor's to adopt, or a later loss
                    break                             # evaluation has overwritten it
                out.append(arr)
            if out is not None:
                return out
        return generic(g_in, homes)
    logits._head_pre = True
    logits._fused_vjp = fused_vjp

    def d_logits(g):
        g = da.asarray(g)
        if g._hv is not None and float(g._hv) == 1.0:
            return dz
        return g * dz
    return build_unary_ops_tensor(logits, d_logits, loss)


_HEAD_FITS = {}


def _head_fits(m, hdim, c):
    key = (m, hdim, c)
    if key not in _HEAD_FITS:
        import ctypes
        fits = ctypes.c_int(0)
        _lib.get().mlp_head_fits(m, hdim, c, _lib.F32, ctypes.byref(fits))
        _HEAD_FITS[key] = bool(fits.value)
    return _HEAD_FITS[key]


def sigmoid_(ts):
    """1 / (1 + exp(-x)) in one kernel; vjp g * s * (1 - s) (closed form; the reference's Sigmoid raises,
    SURVEY F7)."""
    s = da.sigmoid(ts.values)
    return build_unary_ops_tensor(ts, lambda g: g * (s * (1.0 - s)), s)


def softmax_nll_(logits, labels, comm=None):
    """core/losses.py:24-32 as one node: whole-batch max / sum-exp, loss and dz = p - (e*y/q)/m.

    With a communicator the {max, sum-exp} pair of every shard is all-gathered and merged, m is the GLOBAL
    batch size, and the returned loss is this shard's share (the shares add up to the global loss)."""
    if comm is None or comm.world == 1:
        fused = _softmax_head(logits, labels)
        if fused is not None:
            return fused
    z, y = logits.values, as_tensor(labels).values
    dt = z.dtype if z.dtype.kind == "f" else da.get_default_float()
    z, y = z._as_float(dt)._contig(), y._as_float(dt)._contig()
    if z.ndim != 2 or y.shape != z.shape:
        raise ValueError("softmax_nll_: logits %s and labels %s must be equal 2-D shapes" % (z.shape, y.shape))
    m, c = z.shape
    lib = _lib.get()
    stats = da.empty((2,), dt)
    loss = da.empty((), dt)
    dz = da.empty((m, c), dt)
    if comm is None or comm.world == 1:
        # one launch (stats + loss + dz); batches that do not fit one workgroup take the multi-launch form inside
        lib.softmax_nll_fused(z._ptr, y._ptr, m, c, stats._ptr, loss._ptr, dz._ptr, z._code())
    else:
        lib.softmax_nll_stats(z._ptr, m, c, stats._ptr, stats._code())
        stats = comm.merge_softmax_stats(stats)
        lib.softmax_nll_fwd_bwd(z._ptr, y._ptr, m, c, m * comm.world, stats._ptr, loss._ptr, dz._ptr, z._code())
    def d_logits(g):
        g = da.asarray(g)
        if g._hv is not None and float(g._hv) == 1.0:       # the default seed of loss.backward(): no scaling launch
            return dz
        return g * dz
    return build_unary_ops_tensor(logits, d_logits, loss)


# ---------------------------------------------------------------------- anything-in wrappers
def max(obj, axis=None):
    return max_(as_tensor(obj), axis=axis)


def min(obj, axis=None):
    """not in the reference's wrapper list (it only has min_); added for symmetry with max"""
    return min_(as_tensor(obj), axis=axis)


def maximum(obj1, obj2):
    return maximum_(as_tensor(obj1), as_tensor(obj2))


def minimum(obj1, obj2):
    return minimum_(as_tensor(obj1), as_tensor(obj2))


def exp(obj):
    return exp_(as_tensor(obj))


def sum(obj, axis=None):
    return sum_(as_tensor(obj), axis=axis)


def log(obj):
    return log_(as_tensor(obj))


def reshape(obj, newshape):
    return reshape_(as_tensor(obj), newshape)


def pad(obj, pad_width, mode="constant"):
    return pad_(as_tensor(obj), pad_width, mode=mode)


def flatten(obj):
    return flatten_(as_tensor(obj))


def clip(obj, min=None, max=None):
    return clip_(as_tensor(obj), min, max)
