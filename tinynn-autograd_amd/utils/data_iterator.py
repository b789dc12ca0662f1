"""Mini-batch iteration over HBM-resident data (reference protocol: utils/data_iterator.py:8-34).

One call = one epoch: a permutation from `np.random.shuffle` on numpy's GLOBAL RNG (so a seeded run shuffles like
the reference), ONE gather of the whole dataset by that permutation, then consecutive row windows; the final
window may be short.  On device Tensors the gather is a single tnn_gather_rows launch per array and every batch
is a zero-copy view into the gathered buffer — the per-step host work is a Python slice.

`reuse_buffers=True` (new; the reference allocates a fresh `inputs[idx]` every epoch): the gather lands in ONE persistent
pair of epoch buffers (`np.take(..., out=)`), so the batches of every epoch are the SAME views at the same HBM addresses
with new contents — a hipGraph captured over one epoch's batches can be replayed for every later epoch
(examples/mnist_run.py).  The price: a batch object kept from an earlier epoch shows the current epoch's rows.
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

    def __init__(self, batch_size=32, shuffle=True, reuse_buffers=False):
        self.batch_size, self.shuffle, self.reuse_buffers = batch_size, shuffle, reuse_buffers
        self._epoch_buffers = None       # (key, gathered inputs, gathered targets, the batch views) of reuse_buffers
        self.buffers_token = 0           # changes whenever the batches handed out live at NEW addresses
        self._next_order = None          # a permutation drawn ahead of its epoch (prefetch_order)

    def num_batches(self, n_rows):
        return -(-int(n_rows) // self.batch_size)

    def _epoch_order(self, n_rows):
        """The epoch's row permutation (None = natural order).  Exactly one np.random.shuffle call on the global RNG,
        like utils/data_iterator.py:25-26."""
        if not self.shuffle:
            return None
        order, self._next_order = self._next_order, None
        if order is None or len(order) != n_rows:
            order = np.arange(n_rows)
            np.random.shuffle(order)
        return order

    def prefetch_order(self, n_rows):
        """Draw the NEXT epoch's permutation now (new; the reference draws it when the epoch starts).  Same call on the same
        global RNG, so a seeded run is unchanged as long as nothing else draws in between — the caller's promise.  It lets
        the ~0.5 ms host-side shuffle of 50,000 indices run while the GPU is still busy with the current epoch's steps
        (examples/mnist_run.py: right after the epoch's graph has been launched, before the loss read-back waits for it)."""
        if self.shuffle and self._next_order is None:
            order = np.arange(n_rows)
            np.random.shuffle(order)
            self._next_order = order

    def __call__(self, inputs, targets):
        n_rows = len(inputs)
        order = self._epoch_order(n_rows)
        if order is not None and self.reuse_buffers:
            yield from self._reused_epoch(inputs, targets, order)
            return
        self.buffers_token += 1                                    # fresh arrays (or the callers' own) every epoch
        if order is not None:
            inputs, targets = inputs[order], targets[order]        # ops.getitem_ -> row-gather kernel
        for window in range(self.num_batches(n_rows)):
            rows = slice(window * self.batch_size, (window + 1) * self.batch_size)
            yield Batch(inputs=inputs[rows], targets=targets[rows])

    def _reused_epoch(self, inputs, targets, order):
        """One epoch through the persistent buffers: the same gather, written over the previous epoch's rows; the Batch
        objects (and the arrays behind them) are created once and yielded again every epoch."""
        # the cached Batch views were cut with the batch size of their first epoch: a changed `batch_size` attribute rebuilds
        # them (and bumps buffers_token, so a captured epoch graph is not taken for valid).  A DIFFERENT dataset of the same
        # type / shape / dtype is gathered into the same buffers on purpose (that is what "persistent" means here).
        key = (self.batch_size,) + tuple((type(a), getattr(a, "shape", None), str(getattr(getattr(a, "values", a), "dtype", "")))
                                         for a in (inputs, targets))
        if self._epoch_buffers is None or self._epoch_buffers[0] != key:
            gx, gy = inputs[order], targets[order]                 # first epoch: the ordinary gather allocates them
            n_rows = len(gx)
            batches = []
            for window in range(self.num_batches(n_rows)):
                rows = slice(window * self.batch_size, (window + 1) * self.batch_size)
                batches.append(Batch(inputs=gx[rows], targets=gy[rows]))
            self._epoch_buffers = (key, gx, gy, batches)
            self.buffers_token += 1
        else:
            _, gx, gy, batches = self._epoch_buffers
            for src, dst in ((inputs, gx), (targets, gy)):
                np.take(getattr(src, "values", src), order, axis=0, out=getattr(dst, "values", dst))
        for batch in batches:
            yield batch
