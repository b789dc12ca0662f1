"""Create / capture / run / destroy churn on the GPU box: trainers (with their hipGraphs) and op-level models built and dropped
in a loop — the native pool's live bytes must return to where they started and the process must not accumulate device
allocations; also alternating batch sizes on one trainer (row-block step <-> 128-row step)."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, device_array as da

def stats():
    _lib.synchronize(); gc.collect(); da.trim_cache()
    return _lib.pool_stats()

r = bench.FusedRun(bench.WIDTHS_A, 128, "softmax_nll", 4, use_graph=True); r.run(0, 8); del r
s0 = stats()
for i in range(60):
    rows = (128, 256, 512, 96)[i % 4]
    r = bench.FusedRun(bench.WIDTHS_A, rows, "softmax_nll", 4, use_graph=True)
    r.run(0, 8)
    del r
s1 = stats()
print("60 trainers (rows 128 / 256 / 512 / 96, captured, replayed, destroyed): live bytes %d -> %d, pool-cached bytes %d -> %d, device allocations %d -> %d"
      % (s0["live_bytes"], s1["live_bytes"], s0["cached_bytes"], s1["cached_bytes"], s0["device_allocs"], s1["device_allocs"]))
for i in range(30):
    r = bench.OpsRun(bench.WIDTHS_A, 128, "softmax_nll", 4, graph=(i % 2 == 0))
    r.run(0, 8)
    del r
s2 = stats()
print("30 op-level models (eager / captured alternating): live bytes %d -> %d, pool-cached bytes %d -> %d, device allocations %d -> %d"
      % (s1["live_bytes"], s2["live_bytes"], s1["cached_bytes"], s2["cached_bytes"], s1["device_allocs"], s2["device_allocs"]))
# one trainer, alternating batch sizes: every form of the step on the same buffers
rs = np.random.RandomState(0)
trainer = tn.trainer_from_net(bench.build_net(bench.WIDTHS_A), max_rows=1024, loss="softmax_nll", optimizer="adam", lr=1e-3, use_graph=False)
ref = tn.trainer_from_net(bench.build_net(bench.WIDTHS_A), max_rows=1024, loss="softmax_nll", optimizer="adam", lr=1e-3, use_graph=False)
worst = 0.0
for i in range(200):
    rows = int(rs.choice([37, 128, 129, 200, 256, 300, 512, 777, 1024]))
    x = (rs.rand(rows, 784) * (rs.rand(rows, 784) < 0.2)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rs.randint(0, 10, rows)]
    a = float(trainer.step(tn.asarray(x), tn.asarray(y)))
    b = float(ref.step(tn.asarray(x), tn.asarray(y)))
    worst = max(worst, abs(a - b))
same = np.array_equal(np.asarray(trainer.params), np.asarray(ref.params))
print("200 steps at random batch sizes (37 .. 1024) on one trainer: two trainers bit-identical %s, max loss difference %.3g" % (same, worst))
