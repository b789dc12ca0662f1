"""A/B of the fp32 GEMM's LDS-DMA operand path against the register-staged one on config C's five products, interleaved
rounds in one process (the chip's clocks drift with load: variants are compared inside one sustained state).  GPU box only.
Needs a library built with the temporary TNN_GEMM_DMA switch (round 4, while the path was being measured)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd import device_array as da
lib = _lib.get()
rs = np.random.RandomState(0)
SHAPES = [("NN fwd", 0, 0, 512, 4096, 4096), ("NT dX", 0, 1, 512, 4096, 4096), ("TN dW", 1, 0, 4096, 4096, 512),
          ("NN 4096^3", 0, 0, 4096, 4096, 4096)]
ops = []
for name, ta, tb, M, N, K in SHAPES:
    a = da.asarray(rs.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32))
    b = da.asarray(rs.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32))
    c0, c1 = da.empty((M, N), np.float32), da.empty((M, N), np.float32)
    ops.append((name, ta, tb, M, N, K, a, b, c0, c1))
def call(op, dma, out):
    name, ta, tb, M, N, K, a, b, c0, c1 = op
    os.environ["TNN_GEMM_DMA"] = str(dma)
    lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, (M if ta else K), b._ptr, (K if tb else N), 0.0, out._ptr, N, _lib.F32)
for op in ops:      # correctness: the two paths agree to fp32 summation order (they contract k in the same order: bit-equal)
    call(op, 0, op[8]); call(op, 1, op[9])
    x, y = np.asarray(op[8]), np.asarray(op[9])
    print("%-10s max |dma - reg| = %.3g (|c| max %.3g), bit-equal: %s" % (op[0], np.abs(x - y).max(), np.abs(x).max(), np.array_equal(x, y)))
res = {(op[0], d): [] for op in ops for d in (0, 1)}
for rnd in range(8):
    for op in ops:
        for dma in (0, 1):
            e0, e1 = _lib.Event(), _lib.Event()
            reps = 10
            e0.record()
            for _ in range(reps):
                call(op, dma, op[8])
            e1.record()
            if rnd:
                res[(op[0], dma)].append(e0.elapsed_ms(e1) / reps * 1e3)
for op in ops:
    fl = 2.0 * op[3] * op[4] * op[5]
    for dma in (0, 1):
        v = sorted(res[(op[0], dma)])
        med = v[len(v) // 2]
        print("%-10s %s  median %7.1f us (min %7.1f max %7.1f)  %6.1f TFLOP/s" % (op[0], "DMA" if dma else "reg", med, v[0], v[-1], fl / med / 1e6))
