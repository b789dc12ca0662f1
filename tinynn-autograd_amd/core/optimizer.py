"""Optimizers (reference: core/optimizer.py:6-164).

Same contract as the reference: `compute_step(grads, params)` flattens every gradient in layer order
(dict order "w", "b"), calls `_compute_step(flat)` and hands back per-parameter steps that the caller
ADDS to the parameters (core/model.py:59-61).  All state lives in HBM.

  * When the gradients are views into one flat arena (Model binds them that way) the flatten is a
    zero-copy view (core/optimizer.py:14-15 is a concatenate copy in the reference).
  * `Adam` / `SGD` compute the step with ONE fused kernel (tnn_adam with step_out / tnn_ewise_scalar)
    instead of 13 / 1 array expressions; `fused=False` evaluates the reference's literal expressions
    through DeviceArray arithmetic — the two are parity-tested against each other and the oracle.
  * Adam's epsilon is added OUTSIDE the square root, after bias correction (core/optimizer.py:77).
"""

import ctypes
import weakref

import numpy as np

from .. import _lib
from .. import device_array as da


class BaseOptimizer(object):

    def __init__(self, lr, weight_decay):
        self.lr = lr
        self.weight_decay = weight_decay      # stored, unused: the reference's use is commented out (:28-29)
        self._last_flat_step = None

    def compute_step(self, grads, params):
        flat_grad = np.concatenate([np.ravel(g) for layer in grads for g in layer.values()])
        flat_step = self._compute_step(flat_grad)
        self._last_flat_step = flat_step
        steps, offset = [], 0
        for layer in params:
            restored = {}
            for key, p in layer.items():
                count = int(np.prod(p.shape))
                restored[key] = flat_step[offset:offset + count].reshape(p.shape)
                offset += count
            steps.append(restored)
        return steps

    def _compute_step(self, grad):
        raise NotImplementedError

    def apply_flat(self, params, grads):
        """Optional fast path for `Model.step` when parameters and gradients live in flat arenas: update `params`
        in place from `grads` in ONE pass (same maths as compute_step followed by `param += step`,
        core/model.py:56-61) and return True; return False to take the compute_step route."""
        return False


class SGD(BaseOptimizer):
    """step = -lr * g (core/optimizer.py:46-47)"""

    def __init__(self, lr, weight_decay=0.0):
        super().__init__(lr, weight_decay)

    def _compute_step(self, grad):
        return -self.lr * grad

    def apply_flat(self, params, grads):
        if not _flat_pair_ok(params, grads):
            return False
        _lib.get().sgd(params._ptr, grads._ptr, grads.size, self.lr, grads._code())
        return True


_TICKED = weakref.WeakSet()       # Adam instances whose powers a loss launch has advanced for a step that has not run yet


def take_capture_ticks():
    """End of a hipGraph capture (graph.py): the optimizers whose powers a loss launch INSIDE the captured function advances
    while the step that consumes the advance is NOT part of it.  Nothing has executed yet, so their `_ticked` flags are
    cleared here; the captured function sets them after every replay (the replay's loss launch has then really advanced the
    powers and the eager step that follows must not advance them again — exact, no extra launch in the graph).  A captured
    WHOLE step has consumed its tick: nothing to hand over."""
    # the capture is closed: the eager advances its own launches superseded are taken back NOW (nothing of the captured function
    # has executed, so the order "eager advance, roll back, replay advances" is what the device sees)
    for opt in _ROLLBACK:
        if opt._pows is not None:
            _lib.get().adam_tick(opt._pows._ptr, 1.0 / opt._b1, 1.0 / opt._b2)
    del _ROLLBACK[:]
    out = [opt for opt in list(_TICKED) if opt._ticked and opt._tick_captured and opt._pows is not None]
    for opt in out:
        opt._ticked = opt._tick_captured = False
        _TICKED.discard(opt)
    return out


_ROLLBACK = []                    # Adam instances whose pending EAGER advance was superseded inside the open capture


def settle_eager_ticks():
    """Start of a hipGraph capture (graph.py).  An advance that an EAGER loss launch has made and no step has consumed yet (a
    warm-up call of a loss-only function, ANOTHER model's pending loss) is NOT touched here: only an optimizer that ticks or
    steps again inside the captured function has its eager advance taken back — lazily, see `Adam._supersede_eager_tick` and
    `take_capture_ticks` — so capturing one model never perturbs the bias-correction powers of an unrelated one."""
    del _ROLLBACK[:]              # (left over from an aborted capture)


def _flat_pair_ok(params, grads):
    return (isinstance(params, da.DeviceArray) and isinstance(grads, da.DeviceArray) and params.shape == grads.shape
            and params.dtype == grads.dtype and params.dtype.kind == "f" and params.ndim == 1
            and not params._t and not grads._t and params._hv is None and grads._hv is None)


class Adam(BaseOptimizer):
    """reference: core/optimizer.py:50-79"""

    def __init__(self, lr=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, weight_decay=0.0, fused=True):
        super().__init__(lr, weight_decay)
        self._b1, self._b2, self._eps = beta1, beta2, epsilon
        self.fused = fused
        self._t = 0
        self._m = 0
        self._v = 0
        self._pows = None      # device double[4]: {b1^(t-1), b2^(t-1), ticket, pad}
        self._ticked = False   # the powers have ALREADY been advanced for the coming step (by the loss launch, see take_tick)
        self._tick_captured = False   # ... by a launch recorded into an open hipGraph capture (not executed yet)

    def _compute_step(self, grad):
        self._t += 1
        if self.fused and isinstance(grad, da.DeviceArray) and grad.dtype.kind == "f":
            return self._fused_step(grad._contig())
        self._m += (1.0 - self._b1) * (grad - self._m)
        self._v += (1.0 - self._b2) * (grad ** 2 - self._v)
        m_hat = self._m / (1 - self._b1 ** self._t)
        v_hat = self._v / (1 - self._b2 ** self._t)
        return -self.lr * m_hat / (v_hat ** 0.5 + self._eps)

    def _supersede_eager_tick(self):
        """Inside an open hipGraph capture with an EAGER advance of the powers still pending (a loss evaluated before the capture
        whose step never ran): the captured function must record its OWN advance — a replay cannot live on one that happened
        once, outside the graph — so the eager one is forgotten here and taken back when the capture closes
        (`take_capture_ticks`; 1 / b is not the exact inverse in float64: an ulp at most, on this optimizer only)."""
        if self._ticked and not self._tick_captured and _lib.capturing:
            self._ticked = False
            _ROLLBACK.append(self)

    def apply_flat(self, params, grads):
        if not (self.fused and _flat_pair_ok(params, grads)):
            return False
        self._t += 1
        self._ensure_state(grads)
        self._supersede_eager_tick()
        if self._ticked:                             # the loss launch advanced the powers for this step
            _lib.get().adam_ex(params._ptr, grads._ptr, self._m._ptr, self._v._ptr, grads.size, self.lr, self._b1, self._b2,
                               self._eps, self._pows._ptr, None, grads._code(), 0, None, None)
            self._ticked = False
        else:
            _lib.get().adam(params._ptr, grads._ptr, self._m._ptr, self._v._ptr, grads.size, self.lr, self._b1, self._b2,
                            self._eps, self._pows._ptr, None, grads._code())
        return True

    def take_tick(self):
        """For the loss launch of a training step (ops._softmax_head): (powers pointer, b1, b2) if that launch should advance
        the bias-correction powers for the coming step — once per optimizer step, however many losses are evaluated in
        between — else None.  The whole-step trainer does the same in its head launch; it is what lets the step's LAST launch
        carry the optimizer (apply_with_first_layer) without a launch or an arrival counter for the advance."""
        if self.fused and self._pows is not None:
            self._supersede_eager_tick()
        if not self.fused or self._pows is None or self._ticked:
            return None
        self._ticked = True
        self._tick_captured = bool(_lib.capturing)
        _TICKED.add(self)
        return self._pows._ptr, self._b1, self._b2

    def untick(self):
        """Take back an advance of the powers that no step has consumed (a loss-only hipGraph replayed twice in a row: its loss
        launch advances unconditionally).  The multiplication by 1 / b is not the exact inverse in float64 — an ulp at most,
        on this rare path only."""
        if self._ticked and self._pows is not None:
            _lib.get().adam_tick(self._pows._ptr, 1.0 / self._b1, 1.0 / self._b2)
        self._ticked = False

    def apply_with_first_layer(self, params, grads, rows, n_in, n_out, x, dz):
        """The backward of the model's FIRST Dense layer (deferred by Model, core/model.py) and the Adam step over the whole
        arena in ONE launch (tnn_dense_bwd_first_adam): dW = x^T dz and db = column sums of dz are written to the head of
        `grads` (the layer's arena views) and consumed in the same launch.  Returns False if this optimizer cannot."""
        if not (self.fused and _flat_pair_ok(params, grads)) or grads.dtype != np.float32:
            return False
        self._ensure_state(grads)
        self._supersede_eager_tick()
        lib = _lib.get()
        if not self._ticked:                         # no loss launch advanced the powers: a launch of its own
            lib.adam_tick(self._pows._ptr, self._b1, self._b2)
        self._t += 1
        self._ticked = False
        esz = grads.dtype.itemsize
        nw, nb = n_in * n_out, n_out
        rest = nw + nb
        pp, gp, mp, vp = params._ptr, grads._ptr, self._m._ptr, self._v._ptr
        lib.dense_bwd_first_adam(rows, n_in, n_out, x._ptr, dz._ptr, gp, gp + nw * esz, pp, mp, vp, pp + nw * esz,
                                 mp + nw * esz, vp + nw * esz, pp + rest * esz, gp + rest * esz, mp + rest * esz,
                                 vp + rest * esz, grads.size - rest, self.lr, self._b1, self._b2, self._eps,
                                 self._pows._ptr, grads._code())
        return True

    def _ensure_state(self, grad):
        if self._pows is None:
            self._m = da.zeros(grad.shape, grad.dtype)
            self._v = da.zeros(grad.shape, grad.dtype)
            self._pows = da.asarray(np.array([1.0, 1.0, 0.0, 0.0]), dtype=np.float64)

    def _fused_step(self, grad):
        lib = _lib.get()
        self._ensure_state(grad)
        step = da.empty(grad.shape, grad.dtype)
        self._supersede_eager_tick()
        if self._ticked:                             # see take_tick
            lib.adam_ex(None, grad._ptr, self._m._ptr, self._v._ptr, grad.size, self.lr, self._b1, self._b2,
                        self._eps, self._pows._ptr, step._ptr, grad._code(), 0, None, None)
            self._ticked = False
        else:
            lib.adam(None, grad._ptr, self._m._ptr, self._v._ptr, grad.size, self.lr, self._b1, self._b2,
                     self._eps, self._pows._ptr, step._ptr, grad._code())
        return step


class _FusedStateOptimizer(BaseOptimizer):
    """Momentum / RMSProp / Adagrad / Adadelta: the state vectors live in HBM next to the flat gradient and one
    kernel (tnn_optim_step) reads g + state, writes state + step.  `fused=False`, or a gradient that is not a float
    DeviceArray, takes the reference's array expressions instead (same maths through DeviceArray arithmetic)."""
    _kind = None
    _n_state = 1

    def __init__(self, lr, weight_decay=0.0, fused=True):
        super().__init__(lr, weight_decay)
        self.fused = fused
        self._state = None

    def _hyper(self):               # (a, b, eps) of tnn_optim_step
        raise NotImplementedError

    def _array_step(self, grad):
        raise NotImplementedError

    def _compute_step(self, grad):
        if not (self.fused and isinstance(grad, da.DeviceArray) and grad.dtype.kind == "f"):
            return self._array_step(grad)
        grad = grad._contig()
        step = da.empty(grad.shape, grad.dtype)
        self._launch(None, grad, step)
        return step

    def _launch(self, params, grad, step):
        if self._state is None:
            self._state = [da.zeros(grad.shape, grad.dtype) for _ in range(self._n_state)]
        a, b, eps = self._hyper()
        s2 = self._state[1]._ptr if self._n_state > 1 else None
        _lib.get().optim_step(self._kind, params._ptr if params is not None else None, grad._ptr,
                              self._state[0]._ptr, s2, step._ptr if step is not None else None, grad.size, self.lr,
                              a, b, eps, grad._code())

    def apply_flat(self, params, grads):
        if not (self.fused and _flat_pair_ok(params, grads)):
            return False
        self._launch(params, grads, None)
        return True


class Momentum(_FusedStateOptimizer):
    """acc = momentum * acc + g; step = -lr * acc (core/optimizer.py:111-124)"""
    _kind = _lib.OPT_MOMENTUM

    def __init__(self, lr, momentum=0.9, weight_decay=0.0, fused=True):
        super().__init__(lr, weight_decay, fused)
        self._momentum = momentum
        self._acc = 0

    def _hyper(self):
        return self._momentum, 0.0, 0.0

    def _array_step(self, grad):
        self._acc = self._momentum * self._acc + grad
        return -self.lr * self._acc


class RMSProp(_FusedStateOptimizer):
    """reference: core/optimizer.py:82-108"""
    _kind = _lib.OPT_RMSPROP
    _n_state = 2

    def __init__(self, lr=0.01, decay=0.99, momentum=0.0, epsilon=1e-8, weight_decay=0.0, fused=True):
        super().__init__(lr, weight_decay, fused)
        self._decay, self._momentum, self._eps = decay, momentum, epsilon
        self._ms = 0
        self._mom = 0

    def _hyper(self):
        return self._decay, self._momentum, self._eps

    def _array_step(self, grad):
        self._ms += (1 - self._decay) * (grad ** 2 - self._ms)
        self._mom = self._momentum * self._mom + self.lr * grad / (self._ms + self._eps) ** 0.5
        return -self._mom


class Adagrad(_FusedStateOptimizer):
    """reference: core/optimizer.py:127-142"""
    _kind = _lib.OPT_ADAGRAD

    def __init__(self, lr, weight_decay=0.0, epsilon=1e-8, fused=True):
        super().__init__(lr, weight_decay, fused)
        self._G = 0
        self._eps = epsilon

    def _hyper(self):
        return 0.0, 0.0, self._eps

    def _array_step(self, grad):
        self._G += grad ** 2
        return -(self.lr / (self._G + self._eps) ** 0.5) * grad


class Adadelta(_FusedStateOptimizer):
    """reference: core/optimizer.py:145-164"""
    _kind = _lib.OPT_ADADELTA
    _n_state = 2

    def __init__(self, lr=1.0, weight_decay=0.0, decay=0.9, epsilon=1e-8, fused=True):
        super().__init__(lr, weight_decay, fused)
        self._eps, self._decay = epsilon, decay
        self._Eg = 0
        self._delta = 0

    def _hyper(self):
        return self._decay, 0.0, self._eps

    def _array_step(self, grad):
        self._Eg += (1 - self._decay) * (grad ** 2 - self._Eg)
        std = (self._delta + self._eps) ** 0.5
        delta = grad * (std / (self._Eg + self._eps) ** 0.5)
        self._delta += (1 - self._decay) * (delta ** 2 - self._delta)
        return -self.lr * delta
