"""Seeding of numpy's GLOBAL legacy RNG (reference: utils/seeder.py:6-11).

Everything random on the hot path draws from that one stream on the host — the per-epoch shuffle
(utils/data_iterator.py:26) and the Xavier draws (core/initializer.py:83-86) — so a seeded device run consumes
exactly the numbers the reference would.  Nothing is seeded on the GPU: no kernel draws randoms.
"""

import numpy as np

MAX_SEED = 2 ** 32 - 1      # numpy.random.seed's accepted range


def random_seed(seed):
    """Seed numpy's global RNG; ValueError outside [0, 2**32 - 1] (test/test_utils_seeder.py:7-11)."""
    value = int(seed)
    in_range = 0 <= value <= MAX_SEED
    if not in_range:
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    np.random.seed(value)
    return value
