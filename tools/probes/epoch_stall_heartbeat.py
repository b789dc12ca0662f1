#!/usr/bin/env python3
"""Is the paused epoch a stall of the WHOLE device (or process) or of the one queue the epoch graph runs on?  A second thread keeps
a heartbeat on its own HIP stream (a 4-byte hipMemsetAsync + hipStreamSynchronize, ~10-20 us per beat) while the main thread runs the
reference's loop (trainer path); beats longer than 1 ms are listed against the epochs' wall-clock windows.
HEARTBEAT=0: no second thread (control).   HEARTBEAT=host: the thread only sleeps/loops on the host (no GPU work)."""
import ctypes
import os
import sys
import threading
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from tinynn_autograd_amd import _lib                     # noqa: E402
from tinynn_autograd_amd.examples import mnist_run       # noqa: E402

torch.cuda.set_device(0)
lib = _lib.get()
hip = ctypes.CDLL("libamdhip64.so")
mode = os.environ.get("HEARTBEAT", "1")
beats, stop = [], [False]


def heartbeat():
    stream, ptr = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipSetDevice(0) == 0
    assert hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0          # hipStreamNonBlocking
    assert hip.hipMalloc(ctypes.byref(ptr), 256) == 0
    while not stop[0]:
        t = time.time()
        if mode == "host":
            time.sleep(0)
        else:
            hip.hipMemsetAsync(ptr, 0, 4, stream)
            hip.hipStreamSynchronize(stream)
        dt = time.time() - t
        if dt > 1e-3:
            beats.append((t, dt))
        beats_n[0] += 1


beats_n = [0]
(train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
th = None
if mode != "0":
    th = threading.Thread(target=heartbeat, daemon=True)
    th.start()
    time.sleep(0.05)
out, windows = [], []
for rep in range(2):
    np.random.seed(0)
    stats = []
    mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 4, 128, 1e-3, stats=stats, trainer=True)
    lib.stream_sync()
    out.append(" ".join("%6.2f" % (s["steps"] * 1e3) for s in stats))
    windows += [(rep, e, s["wall"], s["steps"]) for e, s in enumerate(stats)]
stop[0] = True
if th is not None:
    th.join(timeout=5)
paused = any(float(v) > 20 for o in out for v in o.split())
print("HEARTBEAT=%-4s %d beats | run 0: %s | run 1: %s%s" % (mode, beats_n[0], out[0], out[1], "   <-- paused" if paused else ""))
for t, dt in beats:
    where = [(r, e) for r, e, w, _ in windows if w[0] <= t + dt and t <= w[1]]
    print("     beat of %.2f ms at %.4f s  (overlaps %s)" % (dt * 1e3, t - windows[0][2][0], " ".join("run %d epoch %d" % w for w in where) or "no training window"))
for r, e, w, st in windows:
    if st > 0.02:
        print("     paused: run %d epoch %d window %.4f .. %.4f s" % (r, e, w[0] - windows[0][2][0], w[1] - windows[0][2][0]))
