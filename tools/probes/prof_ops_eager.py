"""cProfile of the eager op-level training step on the GPU box (where does the host time of `paths.ops_eager` go)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tinynn_autograd_amd import _lib
r = bench.OpsRun(bench.WIDTHS_A, 128, "softmax_nll", 16, graph=False)
for i in range(50):
    r.eager_step(i)
_lib.synchronize()
n = 2000
t0 = time.perf_counter()
for i in range(n):
    r.eager_step(i)
t1 = time.perf_counter()
_lib.synchronize()
t2 = time.perf_counter()
print("host issue time %.1f us/step, drain %.1f us/step" % ((t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    r.eager_step(i)
pr.disable()
_lib.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
