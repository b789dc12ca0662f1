"""Feed-forward network container (reference: core/nn.py:4-31)."""


class Net(object):

    def __init__(self, layers):
        self.layers = layers
        self._phase = "TRAIN"

    def forward(self, inputs):
        for layer in self.layers:
            inputs = layer.forward(inputs)
        return inputs

    def get_parameters(self):
        return [layer.params for layer in self.layers]

    def set_parameters(self, params):
        for layer, new in zip(self.layers, params):
            assert layer.params.keys() == new.keys()
            for key in layer.params.keys():
                assert layer.params[key].shape == new[key].shape
                layer.params[key] = new[key]

    def get_phase(self):
        return self._phase

    def set_phase(self, phase):
        for layer in self.layers:
            layer.set_phase(phase)
        self._phase = phase
