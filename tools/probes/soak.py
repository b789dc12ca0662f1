"""Soak run on the GPU box: (1) the whole-step trainer replayed from hipGraphs for 200 000 steps at 128 and at 256 rows —
losses stay finite, two trainers fed the same batches stay bit-identical, the native pool does not grow; (2) 100 000 eager
op-level steps — the host-side buffer cache and the native pool reach a steady state (no growth after the first 1000 steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib

for rows in (128, 256):
    runs = [bench.FusedRun(bench.WIDTHS_A, rows, "softmax_nll", 16, use_graph=True) for _ in range(2)]
    _lib.synchronize()
    s0 = _lib.pool_stats()
    t0 = time.perf_counter()
    steps = 0
    target = int(os.environ.get("SOAK_STEPS", "200000"))   # (far beyond that Adam at lr 1e-3 on 16 fixed batches drifts until exp underflows: not a runtime matter)
    chunk = runs[0].n_batches
    last = None
    while steps < target:
        for r in runs:
            last = r.run(steps, chunk)
        steps += chunk
    _lib.synchronize()
    dt = time.perf_counter() - t0
    p0, p1 = (np.asarray(r.trainer.params) for r in runs)
    s1 = _lib.pool_stats()
    loss = float(last)
    print("rows %d: %d steps x 2 trainers in %.1f s (%.2f us per step and trainer), parameters finite %s, replicas bit-identical %s, "
          "device allocations %d -> %d, live bytes %d -> %d, last loss %.6f" % (
              rows, steps, dt, dt / steps / 2 * 1e6, bool(np.isfinite(p0).all()), bool(np.array_equal(p0, p1, equal_nan=True)),
              s0["device_allocs"], s1["device_allocs"], s0["live_bytes"], s1["live_bytes"], loss))
    del runs

r = bench.OpsRun(bench.WIDTHS_A, 128, "softmax_nll", 16, graph=False)
for i in range(1000):
    r.eager_step(i)
_lib.synchronize()
s0 = _lib.pool_stats()
t0 = time.perf_counter()
n = int(os.environ.get("SOAK_EAGER_STEPS", "100000"))
for i in range(n):
    r.eager_step(i)
_lib.synchronize()
dt = time.perf_counter() - t0
s1 = _lib.pool_stats()
print("eager op-level path: %d steps in %.1f s (%.1f us per step), device allocations %d -> %d, live bytes %d -> %d, "
      "host-cached bytes %d -> %d, host modules %s" % (
          n, dt, dt / n * 1e6, s0["device_allocs"], s1["device_allocs"], s0["live_bytes"], s1["live_bytes"],
          s0["host_cached_bytes"], s1["host_cached_bytes"], "compiled" if tn.host_modules_compiled() else "interpreted"))
