"""Layer container with the reference's `Net` API (reference: core/nn.py:4-31) plus what the device side needs:
iteration over parameter tensors in the optimizer's flattening order, parameter counting and the Dense/ReLU
structure query the whole-step trainer uses."""

from .layers import Dense, ReLU


class Net(object):
    """Feed-forward stack.  `forward` threads the input through every layer; parameters are exposed exactly as
    the reference does — a list with one {"w": ..., "b": ...} dict per layer (empty dict for activations)."""

    def __init__(self, layers):
        self.layers = list(layers)
        self._phase = "TRAIN"

    # ------------------------------------------------------------------ reference API
    def forward(self, inputs):
        """Thread the input through the layers (core/nn.py:10-13).  A fused `Dense` directly followed by a `ReLU` runs
        as ONE node (GEMM + bias + clip(., 0) epilogue, ops.dense_(relu=True)); the ReLU layer object is then skipped —
        its `inputs` attribute (cached but never read by the reference, core/layers.py:67) holds the fused output."""
        layers, i, activations = self.layers, 0, inputs
        n = len(layers)
        # TRAIN mode, ... Dense -> ReLU -> Dense(classifier): the hidden layer's launch also emits the classifier's logits as
        # partial sums and the classifier's own GEMM is deferred — the loss node runs forward + loss + backward of the head
        # in one launch (ops.softmax_nll_); in TEST mode, or if anything else asks for the logits first, nothing changes
        head = (self._phase == "TRAIN" and n >= 3 and type(layers[-1]) is Dense and layers[-1].fused
                and type(layers[-2]) is ReLU and type(layers[-3]) is Dense and layers[-3].fused and layers[-1].is_init)
        while i < n:
            layer = layers[i]
            nxt = layers[i + 1] if i + 1 < n else None
            if type(layer) is Dense and layer.fused and type(nxt) is ReLU:
                feeds_head = head and i == n - 3
                activations = layer.forward(activations, relu=True, head_w=layers[-1].params["w"] if feeds_head else None,
                                            head_b=layers[-1].params["b"] if feeds_head else None)
                nxt.inputs = activations
                i += 2
            else:
                activations = layer.forward(activations, lazy=True) if (head and i == n - 1) else layer.forward(activations)
                i += 1
        return activations

    def get_parameters(self):
        return [layer.params for layer in self.layers]

    def set_parameters(self, params):
        """Replace parameter tensors layer by layer; keys and shapes must match (asserts, like core/nn.py:19-23)."""
        assert len(params) == len(self.layers), "one parameter dict per layer expected"
        for position, (layer, incoming) in enumerate(zip(self.layers, params)):
            assert set(layer.params) == set(incoming), "layer %d: parameter names differ" % position
            for name, tensor in incoming.items():
                assert tuple(layer.params[name].shape) == tuple(tensor.shape), \
                    "layer %d, %r: shape %s != %s" % (position, name, tensor.shape, layer.params[name].shape)
            layer.params.update(incoming)

    def get_phase(self):
        return self._phase

    def set_phase(self, phase):
        self._phase = phase
        for layer in self.layers:
            layer.set_phase(phase)

    # ------------------------------------------------------------------ device-side helpers
    def __len__(self):
        return len(self.layers)

    def __iter__(self):
        return iter(self.layers)

    def parameter_tensors(self):
        """Initialised parameter tensors in flatten order: layer by layer, dict order ("w" then "b")."""
        return [p for layer in self.layers for p in layer.params.values() if p is not None]

    def num_parameters(self):
        total = 0
        for p in self.parameter_tensors():
            count = 1
            for extent in p.shape:
                count *= int(extent)
            total += count
        return total
