"""Host time per phase of the eager op-level training step (zero_grad / forward / loss / backward / step), no profiler."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, device_array as da
from tinynn_autograd_amd.core.losses import SoftmaxCrossEntropyLoss
from tinynn_autograd_amd.core.model import Model
from tinynn_autograd_amd.core.optimizer import Adam
from tinynn_autograd_amd.core.tensor import Tensor
x_host, y_host = bench.synth_batches(16, 128, bench.WIDTHS_A, "softmax_nll", 0, 1)
X, Y = da.asarray(x_host), da.asarray(y_host)
tb = [(Tensor(X[i * 128:(i + 1) * 128]), Tensor(Y[i * 128:(i + 1) * 128])) for i in range(16)]
loss_layer = SoftmaxCrossEntropyLoss()
model = Model(net=bench.build_net(bench.WIDTHS_A), loss=loss_layer, optimizer=Adam(lr=1e-3))
acc = [0] * 5
n = 3000
for it in range(n + 100):
    xb, yb = tb[it % 16]
    t0 = time.perf_counter_ns(); model.zero_grad()
    t1 = time.perf_counter_ns(); pred = model.forward(xb)
    t2 = time.perf_counter_ns(); out = loss_layer.loss(pred, yb)
    t3 = time.perf_counter_ns(); out.backward()
    t4 = time.perf_counter_ns(); model.step()
    t5 = time.perf_counter_ns()
    if it >= 100:
        for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
            acc[k] += b - a
_lib.synchronize()
names = ("zero_grad", "forward", "loss", "backward", "step")
print("  ".join("%s %.1f us" % (nm, a / n / 1e3) for nm, a in zip(names, acc)), " total %.1f us" % (sum(acc) / n / 1e3))
