#!/usr/bin/env python3
"""Is the paused epoch a once-per-PROCESS event or does it come back after the GPU has been idle?  One process: the dataset on the
device, then runs of four epochs (trainer path) separated by idle sleeps of IDLES seconds (default "0 0.5 0 1 0 2 0 0.2 0 3")."""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from tinynn_autograd_amd import _lib                     # noqa: E402
from tinynn_autograd_amd.examples import mnist_run       # noqa: E402

torch.cuda.set_device(0)
lib = _lib.get()
idles = [float(v) for v in os.environ.get("IDLES", "0 0.5 0 1 0 2 0 0.2 0 3").split()]
(train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
t_origin = time.time()
for rep, idle in enumerate(idles):
    time.sleep(idle)
    np.random.seed(0)
    stats = []
    mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 4, 128, 1e-3, stats=stats, trainer=True)
    lib.stream_sync()
    steps = [s["steps"] * 1e3 for s in stats]
    print("run %2d after %.1f s idle, t = %6.2f s: capture %6.1f  steps %s%s" % (
        rep, idle, stats[0]["wall"][0] - t_origin, stats[0]["capture"] * 1e3, " ".join("%6.2f" % v for v in steps),
        "   <-- paused" if max(steps) > 20 or stats[0]["capture"] > 0.03 else ""))
