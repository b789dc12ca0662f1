"""Layers (reference: core/layers.py:10-98): Dense with lazy shape inference, ReLU / Sigmoid / Tanh.

`Dense.forward` computes `inputs @ w + b` (core/layers.py:49) as one GEMM with a bias epilogue
(ops.dense_); `fused=False` runs the literal two-op expression.  Parameter dict order is "w" then "b"
(core/layers.py:35) — the optimizer's flattening order depends on it.
"""

from . import ops
from .initializer import XavierUniformInit
from .initializer import ZerosInit


class Layer(object):

    def __init__(self, name):
        self.name = name
        self.params, self.grads = {}, {}
        self.is_training = True

    def forward(self, inputs):
        raise NotImplementedError

    def set_phase(self, phase):
        self.is_training = phase == "TRAIN"


class Dense(Layer):

    def __init__(self, num_out, num_in=None, w_init=XavierUniformInit(), b_init=ZerosInit(), fused=True):
        super().__init__("Linear")
        self.initializers = {"w": w_init, "b": b_init}
        self.shapes = {"w": [num_in, num_out], "b": [1, num_out]}
        self.params = {"w": None, "b": None}
        self.fused = fused
        self.is_init = False
        if num_in is not None:
            self._init_parameters(num_in)
        self.inputs = None

    def forward(self, inputs):
        if not self.is_init:                      # lazy: first batch tells the fan-in
            self._init_parameters(inputs.shape[1])
        self.inputs = inputs
        if self.fused:
            return ops.dense_(inputs, self.params["w"], self.params["b"])
        return inputs @ self.params["w"] + self.params["b"]

    def _init_parameters(self, input_size):
        self.shapes["w"][0] = input_size
        for key in ("w", "b"):                    # RNG draw order: w first (b draws nothing)
            self.params[key] = self.initializers[key](shape=self.shapes[key])
            self.params[key].zero_grad()
        self.is_init = True


class Activation(Layer):

    def __init__(self, name):
        super().__init__(name)
        self.inputs = None

    def forward(self, inputs):
        self.inputs = inputs
        return self.func(inputs)

    def func(self, x):
        raise NotImplementedError


class Sigmoid(Activation):
    """1 / (1 + exp(-x)).  The reference's version raises on a Tensor (core/layers.py:79-80);
    here it is one fused kernel whose vjp is the closed form s (1 - s)."""

    def __init__(self):
        super().__init__("Sigmoid")

    def func(self, x):
        return ops.sigmoid_(x)


class Tanh(Activation):
    """Kept as the reference writes it, (1 - e^-x) / (1 + e^-x) = tanh(x / 2) (core/layers.py:88-89)."""

    def __init__(self):
        super().__init__("Tanh")

    def func(self, x):
        e = ops.exp(-x)
        return (1.0 - e) / (1.0 + e)


class ReLU(Activation):
    """clip(x, 0.0): gradient 1 at x == 0 because the mask is x >= 0 (core/layers.py:97-98, ops.py:338)."""

    def __init__(self):
        super().__init__("ReLU")

    def func(self, x):
        return ops.clip(x, 0.0)
