"""Layer container with the reference's `Net` API (reference: core/nn.py:4-31) plus what the device side needs:
iteration over parameter tensors in the optimizer's flattening order, parameter counting and the Dense/ReLU
structure query the whole-step trainer uses."""

from functools import reduce


class Net(object):
    """Feed-forward stack.  `forward` threads the input through every layer; parameters are exposed exactly as
    the reference does — a list with one {"w": ..., "b": ...} dict per layer (empty dict for activations)."""

    def __init__(self, layers):
        self.layers = list(layers)
        self._phase = "TRAIN"

    # ------------------------------------------------------------------ reference API
    def forward(self, inputs):
        return reduce(lambda activations, layer: layer.forward(activations), self.layers, inputs)

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
