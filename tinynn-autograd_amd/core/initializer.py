"""Parameter initialisers (reference: core/initializer.py:9-143).

Draws stay on the HOST through numpy's global MT19937 stream so that a seeded run consumes exactly the
same random numbers as the reference (draw order: SURVEY §3.3); the result is cast to float32 and
uploaded once (core/initializer.py:17-19).  `tests/golden/host_side.npz` holds the reference's own draws for
every class below; `host_side_callers_match_reference` compares bit for bit.
"""

import numpy as np

from .tensor import Tensor


def get_fans(shape):
    """(fan_in, fan_out): rows / columns of a Dense weight, else receptive field x channels / leading extent
    (core/initializer.py:9-12)."""
    if len(shape) != 2:
        return np.prod(shape[1:]), shape[0]
    rows, cols = shape
    return rows, cols


def _gaussian(mean, std, shape):
    return np.random.normal(loc=mean, scale=std, size=shape)


def _flat(low, high, shape):
    return np.random.uniform(low=low, high=high, size=shape)


class Initializer(object):
    """`initializer(shape)` -> float32 device Tensor that requires grad; subclasses provide the host draw `init`."""

    def init(self, shape):
        raise NotImplementedError

    def __call__(self, shape):
        host_values = self.init(shape)
        return Tensor(host_values, requires_grad=True, dtype=np.float32)


class NormalInit(Initializer):
    """N(mean, std)"""

    def __init__(self, mean=0.0, std=1.0):
        self._moments = (mean, std)

    def init(self, shape):
        return _gaussian(*self._moments, shape)


class TruncatedNormalInit(Initializer):
    """scipy truncnorm on [-2 std, 2 std] standard units around `mean` (core/initializer.py:35-41)"""

    def __init__(self, mean=0.0, std=1.0):
        from scipy.stats import truncnorm
        self._dist = truncnorm(-2 * std, 2 * std, loc=mean, scale=std)

    def init(self, shape):
        return self._dist.rvs(size=shape)


class UniformInit(Initializer):
    """U(a, b)"""

    def __init__(self, a=0.0, b=1.0):
        self._range = (a, b)

    def init(self, shape):
        return _flat(*self._range, shape)


class ConstantInit(Initializer):
    """every element = val"""

    def __init__(self, val):
        self._fill = val

    def init(self, shape):
        return np.full(shape=shape, fill_value=self._fill)


class ZerosInit(ConstantInit):
    """the bias default of Dense (core/layers.py:30)"""

    def __init__(self):
        ConstantInit.__init__(self, 0.0)


class _FanScaled(Initializer):
    """Shared body of the four fan-scaled schemes: `_scale(fan_in, fan_out)` times `gain` is the half-width of a
    uniform draw (UNIFORM = True) or the standard deviation of a zero-mean normal draw."""
    UNIFORM = True

    def __init__(self, gain=1.0):
        self._gain = gain

    def _scale(self, fan_in, fan_out):
        raise NotImplementedError

    def init(self, shape):
        width = self._gain * self._scale(*get_fans(shape))
        return _flat(-width, width, shape) if self.UNIFORM else _gaussian(0.0, width, shape)


class XavierUniformInit(_FanScaled):
    """U(-a, a), a = gain * sqrt(6 / (fan_in + fan_out)) — core/initializer.py:83-86"""

    def _scale(self, fan_in, fan_out):
        return np.sqrt(6.0 / (fan_in + fan_out))


class XavierNormalInit(_FanScaled):
    """N(0, gain * sqrt(2 / (fan_in + fan_out))) — core/initializer.py:103-106"""
    UNIFORM = False

    def _scale(self, fan_in, fan_out):
        return np.sqrt(2.0 / (fan_in + fan_out))


class HeUniformInit(_FanScaled):
    """U(-a, a), a = gain * sqrt(6 / fan_in) — core/initializer.py:121-124"""

    def _scale(self, fan_in, fan_out):
        return np.sqrt(6.0 / fan_in)


class HeNormalInit(_FanScaled):
    """N(0, gain * sqrt(2 / fan_in)) — core/initializer.py:139-142"""
    UNIFORM = False

    def _scale(self, fan_in, fan_out):
        return np.sqrt(2.0 / fan_in)
