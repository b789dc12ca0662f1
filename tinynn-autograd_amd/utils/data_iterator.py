"""Mini-batch iteration over HBM-resident data (reference protocol: utils/data_iterator.py:8-34).

One call = one epoch: a permutation from `np.random.shuffle` on numpy's GLOBAL RNG (so a seeded run shuffles like
the reference), ONE gather of the whole dataset by that permutation, then consecutive row windows; the final
window may be short.  On device Tensors the gather is a single tnn_gather_rows launch per array and every batch
is a zero-copy view into the gathered buffer — the per-step host work is a Python slice.
"""

from collections import namedtuple

import numpy as np

Batch = namedtuple("Batch", ["inputs", "targets"])


class BaseIterator(object):
    """Protocol: `iterator(inputs, targets)` yields `Batch(inputs, targets)` for one epoch."""

    def __call__(self, inputs, targets):
        raise NotImplementedError("%s does not implement one epoch" % type(self).__name__)


class BatchIterator(BaseIterator):
    """Consecutive windows of `batch_size` rows over a per-epoch permutation (`shuffle`) or the natural order."""

    def __init__(self, batch_size=32, shuffle=True):
        self.batch_size, self.shuffle = batch_size, shuffle

    def num_batches(self, n_rows):
        return -(-int(n_rows) // self.batch_size)

    def _epoch_order(self, n_rows):
        """The epoch's row permutation (None = natural order).  Exactly one np.random.shuffle call on the global RNG,
        like utils/data_iterator.py:25-26."""
        if not self.shuffle:
            return None
        order = np.arange(n_rows)
        np.random.shuffle(order)
        return order

    def __call__(self, inputs, targets):
        n_rows = len(inputs)
        order = self._epoch_order(n_rows)
        if order is not None:
            inputs, targets = inputs[order], targets[order]        # ops.getitem_ -> row-gather kernel
        for window in range(self.num_batches(n_rows)):
            rows = slice(window * self.batch_size, (window + 1) * self.batch_size)
            yield Batch(inputs=inputs[rows], targets=targets[rows])
