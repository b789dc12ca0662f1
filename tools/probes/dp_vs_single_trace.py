#!/usr/bin/env python3
"""For a kernel trace: the single-GPU step and the world-1 data-parallel step (peer-to-peer transport) of the MNIST net at `rows`
rows, each as 64-step hipGraphs replayed `reps` times, one after the other (first half of the trace: single; second half: DP).
    rocprofv3 --kernel-trace -d out -o t -- python3 tools/probes/dp_vs_single_trace.py 1024
    python3 tools/step_timeline.py out/t_results.db --frac 0.2 ; ... --frac 0.7"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
import tinynn_autograd_amd as tn  # noqa: E402
from tinynn_autograd_amd import _lib  # noqa: E402
from tinynn_autograd_amd.fused import MLPTrainer  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
os.environ["TNN_FORCE_COMM"] = "1"
widths = [784, 256, 128, 10]
rng = np.random.default_rng(0)
x = tn.asarray((rng.random((rows, 784)) * (rng.random((rows, 784)) < 0.19)).astype(np.float32))
y = tn.asarray(np.eye(10, dtype=np.float32)[rng.integers(0, 10, rows)])
comm = tn.dist.init_from_env()
for c in (None, comm):
    t = MLPTrainer(widths, rows, loss="softmax_nll", optimizer="adam", lr=1e-3, comm=c, force_dp=c is not None)
    g = t.capture_steps([(x, y)] * 64)
    for _ in range(reps):
        g.launch()
    _lib.synchronize()
    del g, t
comm.close()
