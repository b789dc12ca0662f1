"""Whole-step trainer: a Python loop of trainer.step() calls (one C call = 4 launches, no graph) against the replay of a captured
graph of the same steps, 391 steps over 391 resident batches; and what capturing + instantiating that graph costs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import bench
from tinynn_autograd_amd import _lib

n = 391
run = bench.FusedRun(bench.WIDTHS_A, 128, "softmax_nll", n, use_graph=False)
tr = run.trainer
for i in range(20):
    run.eager_step(i)
_lib.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(n):
        tr.step(*run.batches[i])
    t1 = time.perf_counter()
    _lib.synchronize()
    t2 = time.perf_counter()
    print("eager: %d trainer.step() calls issued in %.2f ms (%.1f us each), drained after %.2f ms more -> %.2f us per step"
          % (n, (t1 - t0) * 1e3, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3, (t2 - t0) / n * 1e6))
t0 = time.perf_counter()
g = tr.capture_steps(run.batches)
_lib.synchronize()
t1 = time.perf_counter()
print("capture + instantiate of the %d-step graph: %.1f ms" % (n, (t1 - t0) * 1e3))
for rep in range(3):
    t0 = time.perf_counter()
    g.launch()
    _lib.synchronize()
    t1 = time.perf_counter()
    print("replay: %.2f ms -> %.2f us per step" % ((t1 - t0) * 1e3, (t1 - t0) / n * 1e6))
