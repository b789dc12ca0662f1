"""Loss functions (reference: core/losses.py:8-32).

SoftmaxCrossEntropyLoss reproduces the reference's WHOLE-BATCH normalisation (max and sum-exp over all
m x c logits, core/losses.py:26-27), not a per-row softmax.  `fused=True` (default) runs it as
one node / three kernels; `fused=False` evaluates the literal 12-op expression through the autograd ops.
With a communicator the shards exchange one {max, sum-exp} pair so a data-parallel run equals the
single-process run on the global batch.
"""

import numpy as np

from . import ops


class BaseLoss(object):

    def loss(self, predicted, actual):
        raise NotImplementedError


class SoftmaxCrossEntropyLoss(BaseLoss):

    def __init__(self, weight=None, fused=True, comm=None):
        self._weight = np.asarray(weight) if weight is not None else None
        self.fused = fused
        self.comm = comm

    def loss(self, logits, labels):
        if self._weight is not None:
            # the reference indexes the weight vector with the one-hot float matrix and raises
            # IndexError (core/losses.py:30-31); there is no defined behaviour to reproduce
            raise IndexError("arrays used as indices must be of integer (or boolean) type")
        if self.fused:
            return ops.softmax_nll_(logits, labels, comm=self.comm)
        if self.comm is not None and self.comm.world > 1:
            raise ValueError("the unfused loss has no data-parallel form; use fused=True")
        m = logits.shape[0]
        exps = ops.exp(logits - logits.max())
        p = exps / exps.sum()
        nll = -ops.log((p * labels).sum(1))
        return nll.sum() / m


class SquaredErrorLoss(BaseLoss):
    """((pred - target) ** 2).sum() / m — the loss test/test_autograd.py:119-121 builds from ops and the
    one BASELINE.json's 4096-wide autoencoder config uses."""

    def loss(self, predicted, actual):
        m = predicted.shape[0]
        err = predicted - actual
        return (err ** 2).sum() / m
