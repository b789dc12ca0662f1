"""Parameter initialisers (reference: core/initializer.py:9-143).

Draws stay on the HOST through numpy's global MT19937 stream so that a seeded run consumes exactly the
same random numbers as the reference (draw order: SURVEY §3.3); the result is cast to float32 and
uploaded once (core/initializer.py:17-19).
"""

import numpy as np

from .tensor import Tensor


def get_fans(shape):
    """reference: core/initializer.py:9-12"""
    if len(shape) == 2:
        return shape[0], shape[1]
    return np.prod(shape[1:]), shape[0]


class Initializer(object):

    def __call__(self, shape):
        return Tensor(self.init(shape), requires_grad=True, dtype=np.float32)

    def init(self, shape):
        raise NotImplementedError


class NormalInit(Initializer):

    def __init__(self, mean=0.0, std=1.0):
        self._mean, self._std = mean, std

    def init(self, shape):
        return np.random.normal(loc=self._mean, scale=self._std, size=shape)


class TruncatedNormalInit(Initializer):
    """reference: core/initializer.py:35-41 (scipy truncnorm on [-2 std, 2 std] standard units)"""

    def __init__(self, mean=0.0, std=1.0):
        import scipy.stats as stats
        self._tn = stats.truncnorm(-2 * std, 2 * std, loc=mean, scale=std)

    def init(self, shape):
        return self._tn.rvs(size=shape)


class UniformInit(Initializer):

    def __init__(self, a=0.0, b=1.0):
        self._a, self._b = a, b

    def init(self, shape):
        return np.random.uniform(low=self._a, high=self._b, size=shape)


class ConstantInit(Initializer):

    def __init__(self, val):
        self._val = val

    def init(self, shape):
        return np.full(shape=shape, fill_value=self._val)


class ZerosInit(ConstantInit):

    def __init__(self):
        super().__init__(0.0)


class _ScaledUniform(Initializer):
    def __init__(self, gain=1.0):
        self._gain = gain

    def _limit(self, fan_in, fan_out):
        raise NotImplementedError

    def init(self, shape):
        a = self._gain * self._limit(*get_fans(shape))
        return np.random.uniform(low=-a, high=a, size=shape)


class _ScaledNormal(Initializer):
    def __init__(self, gain=1.0):
        self._gain = gain

    def _std(self, fan_in, fan_out):
        raise NotImplementedError

    def init(self, shape):
        std = self._gain * self._std(*get_fans(shape))
        return np.random.normal(loc=0.0, scale=std, size=shape)


class XavierUniformInit(_ScaledUniform):
    """U(-a, a), a = gain * sqrt(6 / (fan_in + fan_out)) — core/initializer.py:83-86"""

    def _limit(self, fan_in, fan_out):
        return np.sqrt(6.0 / (fan_in + fan_out))


class XavierNormalInit(_ScaledNormal):
    """N(0, gain * sqrt(2 / (fan_in + fan_out))) — core/initializer.py:103-106"""

    def _std(self, fan_in, fan_out):
        return np.sqrt(2.0 / (fan_in + fan_out))


class HeUniformInit(_ScaledUniform):
    """U(-a, a), a = gain * sqrt(6 / fan_in) — core/initializer.py:121-124"""

    def _limit(self, fan_in, fan_out):
        return np.sqrt(6.0 / fan_in)


class HeNormalInit(_ScaledNormal):
    """N(0, gain * sqrt(2 / fan_in)) — core/initializer.py:139-142"""

    def _std(self, fan_in, fan_out):
        return np.sqrt(2.0 / fan_in)
