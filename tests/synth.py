"""Pure-numpy synthetic data shared by oracle/gen_golden.py (which feeds it to the real reference) and the parity
tests (which feed the same arrays to the device path).  No product or oracle imports here."""

import numpy as np

EPOCH_CFG = dict(widths=[784, 256, 128, 10], n_train=1000, n_test=500, batch_size=128, num_ep=2, lr=1e-3,
                 seed=5, data_seed=4321)


def epoch_dataset(cfg=EPOCH_CFG):
    """MNIST-shaped rows with ~19 % non-zero pixels and labels a network can learn: argmax of a fixed
    linear teacher on the mean-centred pixels (centred so the ten classes are balanced).  1000 training rows in
    batches of 128 -> 7 full batches + a ragged one of 104."""
    rs = np.random.RandomState(cfg["data_seed"])
    n_in, n_out = cfg["widths"][0], cfg["widths"][-1]
    teacher = rs.randn(n_in, n_out)

    def make(n):
        x = (rs.rand(n, n_in) * (rs.rand(n, n_in) < 0.19)).astype(np.float32)
        return x, np.argmax((x.astype(np.float64) - 0.095) @ teacher, axis=1).astype(np.int64)
    train_x, train_y = make(cfg["n_train"])
    test_x, test_y = make(cfg["n_test"])
    return train_x, train_y, test_x, test_y


def layer_inputs():
    """Inputs of the activation-layer cases (forward + vjp through the layer object)."""
    rs = np.random.RandomState(99)
    x = rs.randn(7, 5) * 2.0
    x[0, 0], x[1, 1] = 0.0, -0.0
    x[2, 2], x[3, 3] = 30.0, -30.0             # saturation on both sides
    g = rs.randn(7, 5)
    return x, g
