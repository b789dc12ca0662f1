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
        with self._graph:
            self.outputs = fn()          # recorded, not executed
            # work the op-level fusions left for a step that is NOT part of fn (a deferred first-layer backward, an advance
            # of Adam's powers by the loss launch) is settled inside the graph; a captured whole step leaves nothing
            from .core import model as _model, optimizer as _optimizer
            _model.settle_pending()
            _optimizer.settle_ticks()

    def __call__(self):
        self._graph.launch()
        return self.outputs

    replay = __call__


def capture(fn, warmup=2):
    """Capture `fn()` (no arguments; it closes over its fixed-address inputs) after `warmup` eager calls."""
    return CapturedFunction(fn, warmup=warmup)
