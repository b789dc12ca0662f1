"""Timing probe for the multi-workgroup classifier head (csrc/tnn_head.hip): the step's per-launch times through the
trainer, and the head alone with / without the per-tile partial logits (TNN_HEAD_CUT=1..3 stops the kernel after the
logits / statistics / dz for an ablation).  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib
import bench
lib = _lib.get()
big = tn.asarray(np.random.RandomState(1).randn(2048, 2048).astype(np.float32))
for _ in range(50):
    big @ big                                            # clocks up before anything is timed
fr = bench.FusedRun(bench.WIDTHS_A, 128, "softmax_nll", 4)
print("per launch via trainer:", fr.per_launch_us())
rs = np.random.RandomState(0)
logits, dz, stats, loss = tn.empty((128, 10)), tn.empty((128, 10)), tn.empty((2,)), tn.empty(())
dw, db, da = tn.empty((128, 10)), tn.empty((10,)), tn.empty((128, 128))
pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
a_r = tn.asarray(np.abs(rs.randn(128, 128)).astype(np.float32))
w_r = tn.asarray((rs.randn(128, 10) * 0.3).astype(np.float32))
b_r = tn.asarray(rs.randn(10).astype(np.float32))
y_r = tn.asarray(np.eye(10, dtype=np.float32)[rs.randint(0, 10, 128)])
zp = tn.asarray(rs.randn(8, 128, 10).astype(np.float32))
def call(part):
    return lambda: lib.mlp_head_tick(128, 128, 10, a_r._ptr, w_r._ptr, b_r._ptr, y_r._ptr, zp._ptr if part else None, logits._ptr,
                                     dz._ptr, stats._ptr, loss._ptr, dw._ptr, db._ptr, da._ptr, _lib.F32, pows._ptr, 0.9, 0.999)
print("head alone, partial logits given: %.2f us" % bench.events_us(call(True), 200))
if not os.environ.get("TNN_HEAD_CUT"):
    print("head alone, logits by MFMA inside: %.2f us" % bench.events_us(call(False), 200))

# the merged launch of the 4-launch step (head + the hidden layer's backward); TNN_HBW_CUT=1..4 stops its tile blocks
# after the logits / statistics / dz / dz1 panel
x_r = tn.asarray(np.abs(rs.randn(128, 256)).astype(np.float32))
w1_r = tn.asarray((rs.randn(256, 128) * 0.1).astype(np.float32))
dw1, db1, dx = tn.empty((256, 128)), tn.empty((128,)), tn.empty((128, 256))
def merged():
    lib.mlp_head_bwd_tick(128, 256, 128, 10, x_r._ptr, w1_r._ptr, a_r._ptr, w_r._ptr, b_r._ptr, y_r._ptr, zp._ptr, logits._ptr,
                          dz._ptr, stats._ptr, loss._ptr, dw._ptr, db._ptr, dw1._ptr, db1._ptr, dx._ptr, _lib.F32, pows._ptr,
                          0.9, 0.999)
print("head + hidden backward, one launch (TNN_HBW_CUT=%s): %.2f us" % (os.environ.get("TNN_HBW_CUT", "0"), bench.events_us(merged, 200)))
def bwd1():
    lib.dense_bwd(128, 256, 128, x_r._ptr, da._ptr, w1_r._ptr, dw1._ptr, db1._ptr, dx._ptr, x_r._ptr, _lib.F32)
print("hidden backward alone: %.2f us" % bench.events_us(bwd1, 200))
