"""Why a step of the epoch graph (391 steps over 50,000 freshly gathered rows) takes ~22.7 us when the headline's 64-step graph over
64 resident batches takes ~21.3: the same trainer step replayed from ONE graph of n steps over n resident batches, n = 64, 128, 391,
782 (inputs 26 / 51 / 157 / 314 MB: inside / beyond what the memory-side cache keeps between replays), HIP events around 5 replays."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import bench
from tinynn_autograd_amd import _lib

ev0, ev1 = _lib.Event(), _lib.Event()
for n in (64, 128, 391, 782, 64):
    run = bench.FusedRun(bench.WIDTHS_A, 128, "softmax_nll", n)
    for _ in range(3):
        run.chunk.launch()
    res = []
    for _ in range(5):
        ev0.record()
        for _ in range(5):
            run.chunk.launch()
        ev1.record()
        res.append(ev0.elapsed_ms(ev1) / 5 / n * 1e3)
    print("one graph of %4d steps over %4d resident batches (%5.1f MB of inputs): %6.2f us per step (min %.2f max %.2f)"
          % (n, n, n * 128 * 784 * 4 / 1e6, np.median(res), min(res), max(res)))
    del run
