"""Pure-numpy synthetic data shared by oracle/gen_golden.py (which feeds it to the real reference) and the parity
tests (which feed the same arrays to the device path).  No product or oracle imports here."""

import numpy as np

EPOCH_CFG = dict(widths=[784, 256, 128, 10], n_train=1000, n_test=500, n_pool=1500, batch_size=128, num_ep=2, lr=1e-3,
                 seed=5, data_seed=4321, min_margin=5e-3)


def epoch_dataset(cfg=EPOCH_CFG):
    """MNIST-shaped rows with ~19 % non-zero pixels and labels a network can learn: argmax of a fixed
    linear teacher on the mean-centred pixels (centred so the ten classes are balanced).  1000 training rows in
    batches of 128 -> 7 full batches + a ragged one of 104.  The test rows are a POOL of n_pool candidates: the fixture
    (tests/golden/epoch.npz, `test_rows`) names the n_test of them the evaluation uses — rows on which the reference
    model's top-2 logit gap exceeds `min_margin` after both epochs, so that "identical argmax / hit_num" is a statement
    about the computation and not about float32-vs-float64 round-off on a near-tie (Adam's sign-like update makes
    unseen-row logits differ by up to ~1e-4 between the two precisions)."""
    rs = np.random.RandomState(cfg["data_seed"])
    n_in, n_out = cfg["widths"][0], cfg["widths"][-1]
    teacher = rs.randn(n_in, n_out)

    def make(n):
        x = (rs.rand(n, n_in) * (rs.rand(n, n_in) < 0.19)).astype(np.float32)
        return x, np.argmax((x.astype(np.float64) - 0.095) @ teacher, axis=1).astype(np.int64)
    train_x, train_y = make(cfg["n_train"])
    test_x, test_y = make(cfg["n_pool"])
    return train_x, train_y, test_x, test_y


def layer_inputs():
    """Inputs of the activation-layer cases (forward + vjp through the layer object)."""
    rs = np.random.RandomState(99)
    x = rs.randn(7, 5) * 2.0
    x[0, 0], x[1, 1] = 0.0, -0.0
    x[2, 2], x[3, 3] = 30.0, -30.0             # saturation on both sides
    g = rs.randn(7, 5)
    return x, g
