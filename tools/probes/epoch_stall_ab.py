#!/usr/bin/env python3
"""Which PRECONDITION makes the one-off paused epoch of the reference's loop (trainer path) appear?  One mode per process:
    MODE=none            nothing in front (torch imported, torch.cuda.set_device only: its HIP context is never created)
    MODE=torch_ctx       torch.cuda.synchronize() first (torch creates its HIP context / streams / allocator), nothing else
    MODE=E               config E measured first through bench.Clock (which calls torch.cuda.synchronize() in its fences)
    MODE=E_no_torch_ctx  config E measured first with a Clock whose torch.cuda.synchronize is replaced by the library's own device sync
prints the trainer path's per-epoch `steps` times of two runs of four epochs."""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bench         # noqa: E402
from tinynn_autograd_amd import _lib                     # noqa: E402
from tinynn_autograd_amd.examples import mnist_run       # noqa: E402

mode = os.environ.get("MODE", "none")
torch.cuda.set_device(0)
lib = _lib.get()
t_start = time.time()


class _NoTorchCuda(object):
    class cuda(object):
        @staticmethod
        def synchronize():
            _lib.synchronize()


if mode == "torch_ctx":
    torch.cuda.synchronize()
elif mode == "E":
    bench.config_e_object(bench.Clock(torch, None, 1))
elif mode == "E_no_torch_ctx":
    bench.config_e_object(bench.Clock(_NoTorchCuda, None, 1))
t_pre = time.time() - t_start
(train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
out = []
for rep in range(2):
    np.random.seed(0)
    stats = []
    mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 4, 128, 1e-3, stats=stats, trainer=True)
    lib.stream_sync()
    out.append(" ".join("%6.2f" % (s["steps"] * 1e3) for s in stats))
print("MODE=%-15s torch ctx %s  pre %.2f s | run 0: %s | run 1: %s%s" % (
    mode, "yes" if torch.cuda.is_initialized() else "no ", t_pre, out[0], out[1],
    "   <-- paused" if any(float(v) > 20 for o in out for v in o.split()) else ""))
