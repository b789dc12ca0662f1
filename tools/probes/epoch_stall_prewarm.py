#!/usr/bin/env python3
"""(also: N_TRAIN rows in the training set, PREPIN=1 = hipHostRegister of the host dataset before the loop)
Does a small warm-up pass of the same loop absorb the one-off pause?  WARM=rows: one epoch of the trainer path over `rows` rows
first (every lazily created runtime object exists afterwards), then SLEEP seconds idle, then twice four epochs over 50,000 rows."""
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
warm, idle = int(os.environ.get("WARM", "0")), float(os.environ.get("SLEEP", "0"))
n_train = int(os.environ.get("N_TRAIN", "50000"))
(train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=n_train, n_test=10000)
if os.environ.get("PREPIN") == "1":
    # the application registers the host dataset itself (hipHostRegister): the runtime's copy then needs no pin / unpin of its own
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    for a in (train_x, test_x):
        rc = hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 0)
        assert rc == 0, rc
w = ""
if warm:
    stats = []
    np.random.seed(1)
    mnist_run.train(train_x[:warm], train_y[:warm], test_x[:1000], test_y[:1000], [256, 128], int(os.environ.get("WARM_EPOCHS", "1")), 128, 1e-3, stats=stats, trainer=True)
    lib.stream_sync()
    w = "warm-up %s ms" % " ".join("%.2f" % (s["train"] * 1e3) for s in stats)
time.sleep(idle)
out = []
for rep in range(2):
    np.random.seed(0)
    stats = []
    mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 4, 128, 1e-3, stats=stats, trainer=True)
    lib.stream_sync()
    out.append(" ".join("%6.2f" % (s["steps"] * 1e3) for s in stats) + " (capture %.1f)" % (stats[0]["capture"] * 1e3))
print("N_TRAIN=%d (%.1f MiB) PREPIN=%s GPU_PINNED_MIN_XFER_SIZE=%s | " % (n_train, train_x.nbytes / 2.0**20, os.environ.get("PREPIN", "0"), os.environ.get("GPU_PINNED_MIN_XFER_SIZE", "-")), end="")
print("WARM=%-6d SLEEP=%.1f %s | run 0: %s | run 1: %s%s" % (warm, idle, w, out[0], out[1],
      "   <-- paused" if any(float(v) > 20 for o in out for v in o.split("(")[0].split()) else ""))
