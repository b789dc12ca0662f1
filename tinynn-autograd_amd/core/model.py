"""Model = net + loss + optimizer (reference: core/model.py:6-68).

`step()` keeps the reference's sequence — collect `.grad` of every parameter, `optimizer.compute_step`,
`param += step` (core/model.py:45-61) — with two MI355X-side additions:

  * flat arenas: after the parameters exist they are re-homed into ONE contiguous HBM buffer and their
    gradients into another, in the optimizer's flattening order.  Flatten/unflatten become views, the
    update is one axpy over the arena, and the gradient arena is exactly the buffer the data-parallel
    all-reduce needs.
  * data parallelism: with a communicator the gradient arena is all-reduced (SUM — the 1/m of the loss is
    already global) between backward() and compute_step, i.e. between examples/mnist/run.py:82 and :83.

`save`/`load` are working .npz counterparts of the reference's pickle pair, which cannot round-trip
(core/model.py:18-35).
"""

import weakref

import numpy as np

from .. import _lib
from .. import device_array as da
from .layers import Dense


_PENDING = weakref.WeakSet()      # models with a deferred first-layer backward


def settle_pending():
    """End of a hipGraph capture (graph.py): a backward whose first-layer launch is still deferred belongs INTO the graph —
    the step that would have absorbed it is outside the captured function."""
    for model in list(_PENDING):
        if model._pending_first is not None:
            model._run_pending_first()
    _PENDING.clear()


class Model(object):

    def __init__(self, net, loss, optimizer, comm=None, use_arena=True):
        self.net = net
        self.loss = loss
        self.optimizer = optimizer
        self.comm = comm
        self.use_arena = use_arena
        self._phase = "TRAIN"
        self._param_arena = None
        self._grad_arena = None
        self._arena_tensors = None
        self._pending_first = None   # (x, dz, w, rows, n_in, n_out): the first Dense layer's backward, deferred to step()

    def forward(self, inputs):
        return self.net.forward(inputs)

    def get_phase(self):
        return self._phase

    def set_phase(self, phase):
        assert phase in ("TRAIN", "TEST")
        self.net.set_phase(phase)
        self._phase = phase

    # ------------------------------------------------------------------ persistence
    def save(self, path):
        arrays = {}
        for i, layer in enumerate(self.net.get_parameters()):
            for key, p in layer.items():
                if p is not None:
                    arrays["%d.%s" % (i, key)] = np.asarray(p.values)
        np.savez(path, **arrays)
        print("Model saved in %s." % path)

    def load(self, path):
        with np.load(path) as data:
            for i, layer in enumerate(self.net.get_parameters()):
                for key, p in layer.items():
                    name = "%d.%s" % (i, key)
                    if p is None or name not in data:
                        continue
                    if tuple(data[name].shape) != tuple(p.shape):
                        raise ValueError("Incompatible architecture. %s in loaded model and %s in "
                                         "defined model." % (data[name].shape, p.shape))
                    p.values = da.asarray(data[name], dtype=p.values.dtype)
                    p.zero_grad()
        print("Restored model from %s." % path)

    # ------------------------------------------------------------------ arenas
    def _live_params(self):
        return [p for layer in self.net.get_parameters() for p in layer.values() if p is not None]

    def _bind_arenas(self):
        tensors = self._live_params()
        if not tensors or (self._arena_tensors is not None and
                           len(tensors) == len(self._arena_tensors) and
                           all(a is b for a, b in zip(tensors, self._arena_tensors))):
            return
        dtypes = {t.values.dtype for t in tensors}
        if len(dtypes) != 1 or next(iter(dtypes)).kind != "f":
            return                                   # mixed dtypes: stay on the generic path
        dt = next(iter(dtypes))
        total = sum(t.values.size for t in tensors)
        params, grads = da.empty((total,), dt), da.zeros((total,), dt)
        off = 0
        # The FIRST Dense layer's two gradient views can be declared pending (device_array.LazyArray): with a fused Adam on
        # one GPU its backward launch is deferred to step(), where ONE launch computes dW / db and applies the optimizer to
        # the whole arena (tnn_dense_bwd_first_adam — the whole-step trainer's last launch); anything that looks at those
        # gradients first (a read, gradient clipping, a second backward) runs the ordinary backward launch instead.
        first = self._first_dense_params(tensors) if (dt == np.float32 and self.comm is None
                                                      and hasattr(self.optimizer, "apply_with_first_layer")) else ()
        self._pending_first = None
        for t in tensors:
            n = t.values.size
            pv = params[off:off + n].reshape(t.shape)
            if any(t is f for f in first):
                gv = da.LazyArray.view(grads, off, t.shape)
                t._defer_first = weakref.ref(self)
            else:
                gv = grads[off:off + n].reshape(t.shape)
                t._defer_first = None
            # the loss launch of a step may advance Adam's powers for it (ops._softmax_head, Adam.take_tick)
            t._tick_optimizer = weakref.ref(self.optimizer) if (first and hasattr(self.optimizer, "take_tick")) else None
            pv[...] = t.values
            had_grad = t._grad is not None
            if had_grad:
                gv[...] = t._grad
            t._values, t._values_home = pv, pv
            t._grad_home = gv
            if had_grad:
                t._grad, t._grad_shared, t._grad_zero = gv, False, False
            off += n
        self._param_arena, self._grad_arena, self._arena_tensors = params, grads, tensors

    def _first_dense_params(self, tensors):
        """(w, b) of the net's first layer when that is a Dense whose parameters head the arena, else ()."""
        layers = getattr(self.net, "layers", None)
        if not layers or type(layers[0]) is not Dense or not layers[0].fused or len(tensors) < 2:
            return ()
        w, b = layers[0].params.get("w"), layers[0].params.get("b")
        if w is None or b is None or tensors[0] is not w or tensors[1] is not b or w.values.ndim != 2:
            return ()
        return (w, b)

    def _defer_first_backward(self, x, dz, w, dw_home, db_home, rows, n_in, n_out):
        """Called by the first Dense layer's vjp (ops._DenseVjp.fused_vjp) instead of launching: remember the operands, mark
        the two arena views pending.  Returns False to decline (the vjp then launches as usual)."""
        ts = self._arena_tensors
        if ts is None or self.comm is not None or dw_home is not ts[0]._grad_home or db_home is not ts[1]._grad_home:
            return False
        if self._pending_first is not None and (dw_home.pending or db_home.pending):
            return False                             # an earlier backward is still outstanding (not dropped by zero_grad)
        self._pending_first = (x, dz, w, rows, n_in, n_out)
        _PENDING.add(self)
        me = weakref.ref(self)

        def materialise(_arr, me=me):
            model = me()
            if model is not None:
                model._run_pending_first()
        dw_home.defer(materialise)
        db_home.defer(materialise)
        return True

    def _run_pending_first(self):
        """The deferred launch after all: dW and db of the first layer into their arena views (tnn_dense_bwd)."""
        pend, self._pending_first = self._pending_first, None
        ts = self._arena_tensors
        if pend is None or ts is None:
            return
        x, dz, w, rows, n_in, n_out = pend
        dw, db = ts[0]._grad_home, ts[1]._grad_home
        _lib.get().dense_bwd(rows, n_in, n_out, x._ptr, dz._ptr, w._ptr, dw.fulfilled_ptr(), db.fulfilled_ptr(), None, None,
                             dz._code())

    # ------------------------------------------------------------------ the training step
    def step(self):
        params = self.net.get_parameters()
        if self.use_arena:
            self._bind_arenas()
        pend = self._pending_first
        if pend is not None:
            ts = self._arena_tensors
            if (ts is not None and ts[0]._grad_home.pending and ts[1]._grad_home.pending
                    and all(t._grad is t._grad_home for t in ts)):
                x, dz, _w, rows, n_in, n_out = pend
                if self.optimizer.apply_with_first_layer(self._param_arena, self._grad_arena, rows, n_in, n_out, x, dz):
                    # ONE launch: the first layer's backward + the optimizer over the whole arena
                    self._pending_first = None
                    ts[0]._grad_home.drop()
                    ts[1]._grad_home.drop()
                    for t in ts:
                        t.values = t._values_home            # same post-state as `param += step`: grad dropped
                    return
            if self._pending_first is not None:
                if ts is not None and (ts[0]._grad_home.pending or ts[1]._grad_home.pending):
                    self._run_pending_first()
                else:
                    self._pending_first = None       # its gradients were dropped (Tensor.zero_grad): nothing to run
        all_grads = [{k: p.grad for k, p in layer.items()} for layer in params]

        if self.comm is not None and self.comm.world > 1:
            # ONE collective over the flat arena only when every gradient really lives there; a gradient that was
            # assigned from outside (`p.grad = ...`, clipping that replaces the array) is first copied home, and a
            # parameter without a gradient keeps the per-tensor loop — never reduce stale arena bytes in its place
            at_home = self._grad_arena is not None
            if at_home:
                for t in self._arena_tensors:
                    if t._grad is None:
                        at_home = False
                        break
                    if t._grad is not t._grad_home:
                        t._grad_home[...] = t._grad
                        t._grad, t._grad_shared, t._grad_zero = t._grad_home, False, False
            if at_home:
                all_grads = [{k: p.grad for k, p in layer.items()} for layer in params]
                self.comm.allreduce(self._grad_arena)
            else:
                for layer in all_grads:
                    for g in layer.values():
                        self.comm.allreduce(g)
            # op-level updates are ordinary array expressions: nothing on the device ties them to the collective, so a
            # peer-to-peer transport is asked (after a stream sync) whether it timed out BEFORE the update is issued
            if getattr(self.comm, "_p2p", False):
                _lib.synchronize()
                self.comm.check(collective=False)

        if (self._param_arena is not None and self._grad_arena is not None
                and all(t._grad is t._grad_home for t in self._arena_tensors)
                and getattr(self.optimizer, "apply_flat", lambda *_: False)(self._param_arena, self._grad_arena)):
            # every gradient already sits in the flat arena: the optimizer updated the parameter arena in place,
            # one pass (core/optimizer.py `_compute_step` + core/model.py:59-61 `param += step`)
            for t in self._arena_tensors:
                t.values = t._values_home                # same post-state as `param += step`: grad dropped
            return

        steps = self.optimizer.compute_step(all_grads, params)

        flat = getattr(self.optimizer, "_last_flat_step", None)
        if (self._param_arena is not None and isinstance(flat, da.DeviceArray)
                and flat.shape == self._param_arena.shape):
            self._param_arena += flat                    # one axpy over every parameter
            for t in self._arena_tensors:
                t.values = t._values_home                # same post-state as `param += step`: grad dropped
            return
        for step, layer in zip(steps, params):
            for key in layer:
                layer[key] += step[key]

    def zero_grad(self):
        self._pending_first = None                   # (Tensor.zero_grad drops the pending marks of the two arena views)
        for layer in self.net.get_parameters():
            for p in layer.values():
                if p is not None:
                    p.zero_grad()
