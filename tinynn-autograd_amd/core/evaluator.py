"""Evaluators (reference: core/evaluator.py:6-114).  Host-side integer / float summaries; predictions and
targets may be numpy arrays or DeviceArrays (converted once)."""

import numpy as np


class BaseEvaluator(object):

    @classmethod
    def evaluate(cls, predictions, targets):
        raise NotImplementedError("Must specify evaluator.")


class AccEvaluator(BaseEvaluator):
    """reference: core/evaluator.py:13-23 — the integer outputs that must be bit-exact"""

    @classmethod
    def evaluate(cls, predictions, targets):
        predictions, targets = np.asarray(predictions), np.asarray(targets)
        total_num = len(predictions)
        hit_num = int(np.sum(predictions == targets))
        return {"total_num": total_num, "hit_num": hit_num, "accuracy": 1.0 * hit_num / total_num}


class EVEvaluator(BaseEvaluator):
    """1 - Var[y - pred] / Var[y], averaged over outputs with non-zero variance (evaluator.py:54-77)"""

    @classmethod
    def evaluate(cls, predictions, targets):
        predictions, targets = np.asarray(predictions), np.asarray(targets)
        assert predictions.shape == targets.shape
        axis = None if predictions.ndim == 1 else 0
        diff_var = np.atleast_1d(np.var(targets - predictions, axis=axis))
        target_var = np.atleast_1d(np.var(targets, axis=axis))
        keep = np.where(target_var != 0)[0]
        return {"mean_ev": np.mean(1.0 - diff_var[keep] / target_var[keep])}


def _per_sample(fn, predictions, targets):
    predictions, targets = np.asarray(predictions), np.asarray(targets)
    assert predictions.shape == targets.shape
    if predictions.ndim == 1:
        return np.mean(fn(predictions - targets))
    if predictions.ndim == 2:
        return np.mean(np.sum(fn(predictions - targets), axis=1))
    raise ValueError("predision supposes to have 1 or 2 dim.")


class MSEEvaluator(BaseEvaluator):
    @classmethod
    def evaluate(cls, predictions, targets):
        return {"mse": _per_sample(np.square, predictions, targets)}


class MAEEvaluator(BaseEvaluator):
    @classmethod
    def evaluate(cls, predictions, targets):
        return {"mse": _per_sample(np.abs, predictions, targets)}   # key name as in the reference (:107)
