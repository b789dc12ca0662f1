"""TEST INFRASTRUCTURE — numpy restatement of the reference's nn stack on top of ref_autograd.

Dense (core/layers.py:25-57), ReLU (:92-98), Tanh (:83-89), whole-batch SoftmaxCrossEntropyLoss
(core/losses.py:24-32), sum-of-squares loss (test/test_autograd.py:119-121), BaseOptimizer.compute_step / SGD / Adam
(core/optimizer.py:12-35,46-47,67-79), Model.step / zero_grad (core/model.py:45-68), and the loop either side of the
step: BatchIterator (utils/data_iterator.py:22-34), argmax + AccEvaluator (examples/mnist/run.py:87-93,
core/evaluator.py:15-23).  It is also the `cpu_baseline` ("port") that bench.py times on the GPU box's host cores.
"""

import numpy as np

from . import ref_autograd as ra
from .ref_autograd import RefTensor


def xavier_uniform(shape):
    """core/initializer.py:83-86 + :17-19 — global numpy RNG, float64 draw cast to float32"""
    a = np.sqrt(6.0 / (shape[0] + shape[1]))
    return RefTensor(np.random.uniform(low=-a, high=a, size=shape), requires_grad=True, dtype=np.float32)


class Dense(object):
    def __init__(self, num_in, num_out):
        self.params = {"w": xavier_uniform([num_in, num_out]),
                       "b": RefTensor(np.full([1, num_out], 0.0), requires_grad=True, dtype=np.float32)}

    def forward(self, x):
        return x @ self.params["w"] + self.params["b"]     # core/layers.py:49


class ReLU(object):
    params = {}

    def forward(self, x):
        return ra.clip(x, 0.0)                             # core/layers.py:97-98


class Tanh(object):
    params = {}

    def forward(self, x):
        """core/layers.py:88-89 — (1 - e^-x) / (1 + e^-x) = tanh(x / 2), with exp(-x) evaluated twice as there"""
        return (1.0 - ra.exp(-x)) / (1.0 + ra.exp(-x))


def build_mlp(widths):
    """Dense/ReLU stack, eager init in layer order (one RNG draw per weight matrix)."""
    layers = []
    for i in range(len(widths) - 1):
        layers.append(Dense(widths[i], widths[i + 1]))
        if i < len(widths) - 2:
            layers.append(ReLU())
    return layers


def forward(layers, x):
    for layer in layers:
        x = layer.forward(x)
    return x


def softmax_nll(logits, labels):
    """core/losses.py:24-32 — max and sum-exp over the WHOLE batch"""
    m = logits.shape[0]
    exps = ra.exp(logits - logits.max())
    p = exps / exps.sum()
    nll = -ra.log((p * labels).sum(1))
    return nll.sum() / m


def squared_error(pred, target):
    m = pred.shape[0]
    err = pred - target
    return (err ** 2).sum() / m


class SGD(object):
    def __init__(self, lr):
        self.lr = lr

    def flat_step(self, g):
        return -self.lr * g                                # core/optimizer.py:46-47


class Adam(object):
    def __init__(self, lr=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, epsilon
        self.t, self.m, self.v = 0, 0, 0

    def flat_step(self, g):                                # core/optimizer.py:67-79
        self.t += 1
        self.m += (1.0 - self.b1) * (g - self.m)
        self.v += (1.0 - self.b2) * (g ** 2 - self.v)
        m_hat = self.m / (1 - self.b1 ** self.t)
        v_hat = self.v / (1 - self.b2 ** self.t)
        return -self.lr * m_hat / (v_hat ** 0.5 + self.eps)


def parameters(layers):
    return [p for layer in layers for p in layer.params.values()]


def zero_grad(layers):                                     # core/model.py:63-68
    for p in parameters(layers):
        p.zero_grad()


def apply_step(layers, opt):
    """core/model.py:45-61 + core/optimizer.py:12-35: flatten, step, unflatten, `param += step`"""
    ps = parameters(layers)
    flat = np.concatenate([np.ravel(p.grad) for p in ps])
    step = opt.flat_step(flat)
    off = 0
    for p in ps:
        n = int(np.prod(p.shape))
        p += step[off:off + n].reshape(p.shape)
        off += n


def train_step(layers, opt, loss_fn, x, y):
    """examples/mnist/run.py:79-84; returns (loss value, logits array)"""
    zero_grad(layers)
    pred = forward(layers, RefTensor(x))
    loss = loss_fn(pred, RefTensor(y))
    loss.backward()
    apply_step(layers, opt)
    return loss.values, pred.values


def epoch_batches(inputs, targets, batch_size, shuffle=True):
    """utils/data_iterator.py:22-34 — one np.random.shuffle on the GLOBAL RNG per epoch, whole-dataset gather through
    getitem, consecutive row windows (the last one may be short)"""
    starts = np.arange(0, len(inputs), batch_size)
    if shuffle:
        idx = np.arange(len(inputs))
        np.random.shuffle(idx)
        inputs, targets = inputs[idx], targets[idx]
    for start in starts:
        yield inputs[start:start + batch_size], targets[start:start + batch_size]


def acc_evaluate(predictions, targets):
    """core/evaluator.py:15-23"""
    hit = int(np.sum(predictions == targets))
    return {"total_num": len(predictions), "hit_num": hit, "accuracy": 1.0 * hit / len(predictions)}


def train_epochs(widths, train_x, train_y_onehot, test_x, test_y, num_ep, batch_size, lr):
    """examples/mnist/run.py:50-93 with Dense layers that initialise lazily: the RNG is consumed in the reference's
    order — first epoch's shuffle, THEN W0, W1, ... at the first forward (core/layers.py:45-46).
    Returns (per-step losses, per-epoch argmax vectors, per-epoch AccEvaluator dicts)."""
    layers, opt = None, Adam(lr=lr)
    train_x, train_y = RefTensor(train_x), RefTensor(train_y_onehot)
    test_x = RefTensor(test_x)
    losses, preds, results = [], [], []
    for _ in range(num_ep):
        for bx, by in epoch_batches(train_x, train_y, batch_size):
            if layers is None:
                layers = build_mlp(widths)                   # lazy init happens inside the first forward
            zero_grad(layers)
            loss = softmax_nll(forward(layers, bx), by)
            loss.backward()
            apply_step(layers, opt)
            losses.append(float(loss.values))
        idx = np.argmax(forward(layers, test_x).values, axis=1)
        preds.append(idx)
        results.append(acc_evaluate(idx, test_y))
    return losses, preds, results
