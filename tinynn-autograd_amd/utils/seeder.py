"""reference: utils/seeder.py:6-11"""

import numpy as np


def random_seed(seed):
    seed = int(seed)
    if not 0 <= seed <= 2 ** 32 - 1:
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    np.random.seed(seed)
