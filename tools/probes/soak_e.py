"""configs[4] (8192-wide x 4, bf16 storage) for 3000 steps on one GPU: the loss keeps falling and stays finite; the fp32 master
weights, the bf16 working copy and its transpose stay consistent (W16 == bf16(W), W16T == W16^T) at the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, bf16
r = bench.FusedRun(bench.WIDTHS_E, 512, "mse", 2, dtype="bfloat16")
losses = []
t0 = time.perf_counter()
n = int(os.environ.get("SOAK_E_STEPS", "3000"))
for i in range(n):
    l = r.eager_step(i)
    if i % 250 == 0 or i == n - 1:
        losses.append(float(np.asarray(l)))
_lib.synchronize()
dt = time.perf_counter() - t0
print("%d steps in %.1f s (%.3f ms per step); loss every 250 steps: %s" % (n, dt, dt / n * 1e3, ["%.5g" % v for v in losses]))
tr = r.trainer
ok = True
for l in range(tr.n_layers):
    w = np.asarray(tr.param_view(l, "w"))
    w16 = np.asarray(tr.weights_bf16(l))
    ref = np.asarray(bf16.to_bf16(tn.asarray(w)))
    same = np.array_equal(w16, ref)
    print("layer %d: master finite %s, bf16 working copy == bf16(master) %s" % (l, bool(np.isfinite(w).all()), same))
    ok = ok and same and bool(np.isfinite(w).all())
print("OK" if ok and all(np.isfinite(losses)) and losses[-1] < losses[0] else "CHECK")
