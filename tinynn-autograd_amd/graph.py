"""hipGraph capture of op-level code: run a whole training step written against the reference API
(`model.zero_grad(); pred = model.forward(x); loss = loss_layer.loss(pred, y); loss.backward(); model.step()`,
examples/mnist/run.py:79-83) ONCE under stream capture and replay its ~40 kernel launches with a single
hipGraphLaunch — the Python autograd bookkeeping, the ctypes calls and the per-launch host cost disappear from the
steady state (the MNIST-size step is dispatch-bound).

What makes a step capturable
  * its inputs live at FIXED device addresses: keep two staging Tensors and copy each batch into them
    (`x_stage.values[...] = batch.inputs.values`) before replaying;
  * nothing inside reads a value back to the host (`float(loss.values)`, `.tolist()`, `np.asarray`) — hipStreamCapture
    rejects the synchronisation and `capture()` raises;
  * every per-step quantity that changes lives on the DEVICE: this package's `Adam(fused=True)` keeps b1^t / b2^t in
    HBM and advances them in-kernel; an optimizer that bakes a host-side step counter into kernel arguments (the
    reference's own optimizer.py, or `Adam(fused=False)`) would replay step 1's bias correction forever;
  * shapes are static (a ragged last batch needs its own capture or an eager step).
Buffers allocated while capturing belong to the graph (pool allocator tags them), so the tensors returned by the
captured function — e.g. the loss — stay valid and are refreshed by every replay.
"""

from . import _lib


class CapturedFunction(object):

    def __init__(self, fn, warmup=2):
        self._fn = fn
        for _ in range(warmup):          # real steps: lazy Dense init, arena binding, optimizer state creation
            fn()
        _lib.synchronize()
        self._graph = _lib.Graph()
        from . import device_array as _da
        from .core import model as _model, optimizer as _optimizer, tensor as _tensor
        _tensor.take_capture_grads()
        _da.take_capture_lazies()        # (anything left over from an aborted capture)
        # an advance of Adam's powers made by an EAGER loss launch (a warm-up call above, another model's pending loss) that
        # no step has consumed is taken back first: the capture's own loss launch must record the advance in the graph
        _optimizer.settle_eager_ticks()
        with self._graph:
            self.outputs = fn()          # recorded, not executed
            # work the op-level fusions left for a step that is NOT part of fn: a deferred first-layer backward is settled
            # inside the graph; a captured whole step leaves nothing
            _model.settle_pending()
        # ... an advance of Adam's powers by a loss launch whose step runs outside: flagged after every replay;
        # deferred arrays nobody has read yet (TRAIN-phase logits of a captured forward, the hidden gradient of the fused
        # head): re-armed after every replay so that a read sees THAT replay's values
        self._ticked = _optimizer.take_capture_ticks()
        self._pending = _optimizer._TICKED
        self._lazies = _da.take_capture_lazies()
        # ... and the gradient attributes of the parameters the function accumulated into (an eager step between replays drops
        # them): re-installed after every replay
        self._grads = _tensor.take_capture_grads()

    def __call__(self):
        for opt in self._ticked:
            opt.untick()                 # the previous replay's advance was never consumed by a step
        self._graph.launch()
        for ref, thunk in self._lazies:
            arr = ref()
            if arr is not None:
                arr._thunk = thunk
        for ref, grad, shared in self._grads:
            t = ref()
            if t is not None:
                t._grad, t._grad_shared, t._grad_zero = grad, shared, False
        for opt in self._ticked:
            opt._ticked, opt._tick_captured = True, False      # this replay's loss launch HAS advanced the powers
            self._pending.add(opt)
        return self.outputs

    replay = __call__


def capture(fn, warmup=2):
    """Capture `fn()` (no arguments; it closes over its fixed-address inputs) after `warmup` eager calls."""
    return CapturedFunction(fn, warmup=warmup)
