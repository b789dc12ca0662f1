"""TEST INFRASTRUCTURE — closed-form float64 training step of a Dense/ReLU MLP.

What the reference's op graph computes for the hot path, written out:
    z_l = a_{l-1} W_l + b_l ,  a_l = max(z_l, 0)  (mask z >= 0, core/ops.py:338)
    softmax NLL over the whole batch (core/losses.py:24-32):  dz_L = p - (e*y/q)/m  (= p - y/m, one-hot)
    sum of squares / m:                                        dz_L = 2 (pred - y) / m
    dW_l = a_{l-1}^T dz_l ,  db_l = column-sum dz_l ,  dz_{l-1} = (dz_l W_l^T) * [z_{l-1} >= 0]
    Adam / SGD on the flattened (w, b, w, b, ...) vector (core/optimizer.py:14-15,46-47,67-79).
Used where the op-graph oracle (ref_nn) would be too slow (4096-wide layers) and as the fp64 yardstick
for GPU-vs-CPU checks at full size.  Validated against the real reference by gen_golden.py.
"""

import numpy as np


class ClosedFormMLP(object):

    def __init__(self, weights, biases, loss="softmax_nll", optimizer="adam", lr=1e-3, beta1=0.9,
                 beta2=0.999, epsilon=1e-8):
        self.W = [np.array(w, dtype=np.float64) for w in weights]
        self.b = [np.array(b, dtype=np.float64).reshape(1, -1) for b in biases]
        self.loss, self.optimizer = loss, optimizer
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, epsilon
        self.t = 0
        self.m = self.v = 0.0

    def forward(self, x):
        acts, zs = [np.asarray(x, dtype=np.float64)], []
        for l, (w, b) in enumerate(zip(self.W, self.b)):
            z = acts[-1] @ w + b
            zs.append(z)
            acts.append(np.clip(z, 0.0, None) if l < len(self.W) - 1 else z)
        return acts, zs

    def loss_and_grads(self, x, y, m_global=None, masks=None):
        """masks (optional): per hidden layer boolean arrays to use INSTEAD of (z >= 0) — lets a float32 device
        run be checked exactly where ReLU's derivative is discontinuous (a pre-activation within float32
        rounding of zero legitimately lands on the other side)."""
        y = np.asarray(y, dtype=np.float64)
        acts, zs = self.forward(x)
        out = acts[-1]
        m = out.shape[0] if m_global is None else m_global
        if self.loss == "softmax_nll":
            e = np.exp(out - out.max())
            s = e.sum()
            q = (e * y).sum(1, keepdims=True)
            loss = float((np.log(s) - np.log(q)).sum() / m)
            dz = e / s - (e * y / q) / m
        else:
            err = out - y
            loss = float((err ** 2).sum() / m)
            dz = 2.0 * err / m
        gW, gb = [None] * len(self.W), [None] * len(self.W)
        for l in reversed(range(len(self.W))):
            gW[l] = acts[l].T @ dz
            gb[l] = dz.sum(0, keepdims=True)
            if l > 0:
                mask = (zs[l - 1] >= 0) if masks is None else masks[l - 1]
                dz = (dz @ self.W[l].T) * mask
        self.last_pre_activations = zs
        return loss, out, gW, gb

    def step(self, x, y):
        loss, out, gW, gb = self.loss_and_grads(x, y)
        flat = np.concatenate([np.ravel(g) for pair in zip(gW, gb) for g in pair])
        if self.optimizer == "sgd":
            upd = -self.lr * flat
        else:
            self.t += 1
            self.m = self.m + (1.0 - self.b1) * (flat - self.m)
            self.v = self.v + (1.0 - self.b2) * (flat ** 2 - self.v)
            m_hat = self.m / (1 - self.b1 ** self.t)
            v_hat = self.v / (1 - self.b2 ** self.t)
            upd = -self.lr * m_hat / (v_hat ** 0.5 + self.eps)
        off = 0
        for l in range(len(self.W)):
            for arr in (self.W[l], self.b[l]):
                n = arr.size
                arr += upd[off:off + n].reshape(arr.shape)
                off += n
        return loss, out, gW, gb
