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

import numpy as np

from .. import device_array as da


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
        for t in tensors:
            n = t.values.size
            pv = params[off:off + n].reshape(t.shape)
            gv = grads[off:off + n].reshape(t.shape)
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

    # ------------------------------------------------------------------ the training step
    def step(self):
        params = self.net.get_parameters()
        if self.use_arena:
            self._bind_arenas()
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
                from .. import _lib
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
        for layer in self.net.get_parameters():
            for p in layer.values():
                if p is not None:
                    p.zero_grad()
