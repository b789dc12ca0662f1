"""Is the sporadic one-off ~70 ms stall of an early epoch of the reference's loop a full (generation-2) collection of Python's garbage
collector?  The loop as bench.py runs it (torch imported, the dataset on the device), trainer path, with a gc callback that logs every
collection's generation and duration next to the per-epoch times."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (bench.py has it loaded: its objects are part of what a full collection walks)

import bench  # noqa: F401
from tinynn_autograd_amd.examples import mnist_run

events, t_start = [], {}


def cb(phase, info):
    if phase == "start":
        t_start[info["generation"]] = time.perf_counter()
    else:
        events.append((info["generation"], (time.perf_counter() - t_start[info["generation"]]) * 1e3, info["collected"]))


gc.callbacks.append(cb)
(train_x, train_y), (test_x, test_y), _ = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
for rep in range(4):
    np.random.seed(0)
    stats = []
    del events[:]
    mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 6, 128, 1e-3, stats=stats, trainer=True)
    print("run %d  epochs (ms): %s" % (rep, " ".join("%.1f" % (s["train"] * 1e3) for s in stats)))
    print("        collections >= 1 ms: %s" % ", ".join("gen %d %.1f ms (%d objects)" % e for e in events if e[1] >= 1.0))
    print("        tracked objects: %d" % len(gc.get_objects()))
