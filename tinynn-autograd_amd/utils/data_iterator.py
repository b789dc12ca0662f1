"""Mini-batch iterator (reference: utils/data_iterator.py:8-34).

Same protocol: per-epoch `np.random.shuffle(idx)` on numpy's GLOBAL RNG, a full-dataset gather
`inputs[idx]`, then contiguous row slices; the last batch may be ragged.  With device Tensors the gather
is one tnn_gather_rows launch per array and every batch is a zero-copy view.
"""

from collections import namedtuple

import numpy as np

Batch = namedtuple("Batch", ["inputs", "targets"])


class BaseIterator(object):

    def __call__(self, inputs, targets):
        raise NotImplementedError


class BatchIterator(BaseIterator):

    def __init__(self, batch_size=32, shuffle=True):
        self.batch_size = batch_size
        self.shuffle = shuffle

    def __call__(self, inputs, targets):
        n = len(inputs)
        if self.shuffle:
            idx = np.arange(n)
            np.random.shuffle(idx)
            inputs, targets = inputs[idx], targets[idx]
        for start in range(0, n, self.batch_size):
            end = start + self.batch_size
            yield Batch(inputs=inputs[start:end], targets=targets[start:end])
