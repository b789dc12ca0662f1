"""Layers with the reference's API (reference: core/layers.py:10-98): `Dense` with lazy fan-in inference and the
`ReLU` / `Sigmoid` / `Tanh` activations.

Device-side differences: `Dense.forward` issues ONE GEMM with a bias epilogue (ops.dense_) instead of a matmul
node plus a broadcast-add node (`fused=False` restores the literal `inputs @ w + b`, core/layers.py:49);
`Sigmoid` is a single fused kernel (the reference's raises on a Tensor).  The parameter dict order is
"w" then "b" (core/layers.py:35): the optimizer's flatten order and the trainer's arena layout depend on it.
"""

from . import ops
from .initializer import XavierUniformInit
from .initializer import ZerosInit

PARAM_ORDER = ("w", "b")


class Layer(object):
    """Base class: a name, a parameter dict (empty for activations) and the TRAIN/TEST flag nobody reads
    (core/layers.py:21-22)."""

    def __init__(self, name):
        self.name = name
        self.params = {}
        self.grads = {}
        self.is_training = True

    def forward(self, inputs):
        raise NotImplementedError

    def set_phase(self, phase):
        self.is_training = (phase == "TRAIN")


class Dense(Layer):
    """y = x w + b with w: [num_in, num_out], b: [1, num_out].  num_in may be omitted and is then read off the
    first batch (core/layers.py:43-57)."""

    def __init__(self, num_out, num_in=None, w_init=XavierUniformInit(), b_init=ZerosInit(), fused=True):
        super().__init__("Linear")
        self.fused = fused
        self.initializers = dict(zip(PARAM_ORDER, (w_init, b_init)))
        self.shapes = {"w": [num_in, num_out], "b": [1, num_out]}
        self.params = dict.fromkeys(PARAM_ORDER)
        self.inputs = None
        self.is_init = False
        if num_in is not None:
            self._init_parameters(num_in)

    def _init_parameters(self, input_size):
        """Draw the parameters (host RNG, w before b — only w consumes random numbers) and upload them."""
        self.shapes["w"][0] = input_size
        for name in PARAM_ORDER:
            tensor = self.initializers[name](shape=self.shapes[name])
            tensor.zero_grad()
            self.params[name] = tensor
        self.is_init = True

    def forward(self, inputs, relu=False, head_w=None, lazy=False, head_b=None):
        """relu=True is passed by Net.forward when the next layer is a ReLU: one launch for both (ops.dense_); head_w / lazy:
        the classifier-head arrangements of Net.forward (ops.dense_)."""
        if not self.is_init:
            self._init_parameters(inputs.shape[1])
        self.inputs = inputs                     # kept like the reference does (core/layers.py:48)
        w, b = (self.params[name] for name in PARAM_ORDER)
        if not self.fused:
            out = inputs @ w + b
            return ops.clip(out, 0.0) if relu else out
        return ops.dense_(inputs, w, b, relu=relu, head_w=head_w, lazy=lazy, head_b=head_b)


class Activation(Layer):
    """Parameter-free layer applying `func` (core/layers.py:60-72)."""

    def __init__(self, name):
        super().__init__(name)
        self.inputs = None

    def func(self, x):
        raise NotImplementedError

    def forward(self, inputs):
        self.inputs = inputs
        return self.func(inputs)


class ReLU(Activation):
    """clip(x, 0.0): the vjp mask is x >= 0, i.e. gradient 1 AT zero (core/layers.py:97-98, core/ops.py:338)."""

    def __init__(self):
        super().__init__("ReLU")

    def func(self, x):
        return ops.clip(x, 0.0)


class Sigmoid(Activation):
    """1 / (1 + exp(-x)) as one kernel; vjp = s (1 - s)."""

    def __init__(self):
        super().__init__("Sigmoid")

    def func(self, x):
        return ops.sigmoid_(x)


class Tanh(Activation):
    """The reference's formula verbatim in meaning: (1 - e^-x) / (1 + e^-x), which is tanh(x / 2), not tanh(x)
    (core/layers.py:88-89) — reproduced, not corrected."""

    def __init__(self):
        super().__init__("Tanh")

    def func(self, x):
        decay = ops.exp(-x)
        return (1.0 - decay) / (1.0 + decay)
