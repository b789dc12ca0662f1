"""Where the bf16 dW + Adam launch's time goes: K sweep (the optimizer stream with a product of growing size under it), the row
stride of p / m / v (8192 vs 8320 columns), against the stand-alone 2-d optimizer kernel on the same box.  Parameter sets rotate
over three copies (3 x 0.94 GB: nothing survives in the memory-side cache).  Round 5: K = 64 -> 512 adds ~22 us (the product is
almost hidden under the optimizer's stream), the stream itself runs at 0.94 of the stand-alone optimizer's rate, the row stride
does not matter."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, bf16

lib = _lib.get()
rs = np.random.RandomState(3)
ev0, ev1 = _lib.Event(), _lib.Event()
pows = tn.asarray(np.array([0.9, 0.999, 0.0, 0.0]), dtype=np.float64)


def timed(fn, reps=12):
    for i in range(3):
        fn(i)
    ev0.record()
    for i in range(reps):
        fn(i)
    ev1.record()
    return ev0.elapsed_ms(ev1) / reps * 1e3


for W in (8192, 8320):
    sets = [dict(p=tn.asarray(rs.uniform(-0.03, 0.03, (W, W)).astype(np.float32)), m=tn.zeros((W, W), np.float32), v=tn.zeros((W, W), np.float32),
                 w=tn.empty((W, W), np.uint16), wt=tn.empty((W, W), np.uint16)) for _ in range(3)]
    G = tn.asarray(rs.uniform(-1e-3, 1e-3, (W, W)).astype(np.float32))

    def adam(i):
        s = sets[i % 3]
        lib.adam_master_bf16_2d(s["p"]._ptr, G._ptr, s["m"]._ptr, s["v"]._ptr, s["w"]._ptr, s["wt"]._ptr, W, W, 1e-3, 0.9, 0.999, 1e-8, pows._ptr, 0)
    us = timed(adam)
    print("W %d  stand-alone optimizer (32 B/param) %7.1f us  %.2f TB/s" % (W, us, 32.0 * W * W / us / 1e6))
    for K in (64, 128, 256, 512):
        inT = bf16.to_bf16(rs.uniform(0, 1, (W, K)).astype(np.float32))
        dzT = bf16.to_bf16(rs.uniform(-1e-3, 1e-3, (W, K)).astype(np.float32))
        for name, w, wt in (("W and W^T", True, True), ("W only", True, False), ("W^T only", False, True)):
            def call(i):
                s = sets[i % 3]
                lib.gemm_bf16_nt_adam(W, W, K, inT._ptr, K, dzT._ptr, K, None, s["p"]._ptr, s["m"]._ptr, s["v"]._ptr, s["w"]._ptr if w else None,
                                      s["wt"]._ptr if wt else None, 1e-3, 0.9, 0.999, 1e-8, pows._ptr)
            us = timed(call)
            print("W %d  dW + Adam, K %4d, bf16 copies: %-10s %7.1f us  %.2f TB/s" % (W, K, name, us, (24.0 + 2 * w + 2 * wt) * W * W / us / 1e6))
    del sets, G
