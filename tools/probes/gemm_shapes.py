"""Layout vs shape: the fp32 MFMA GEMM on 4096x4096xK for K = 512 / 4096 in every layout (which of the TN dW launch's
properties costs the time — its MN-contiguous operands, its short K, or its 64 MB of output?).  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd import device_array as da
lib = _lib.get()
rs = np.random.RandomState(0)
for (M, N, K) in ((4096, 4096, 512), (4096, 4096, 4096), (512, 4096, 4096)):
    for name, ta, tb in (("NN", 0, 0), ("NT", 0, 1), ("TN", 1, 0), ("TT", 1, 1)):
        a = da.asarray(rs.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32))
        b = da.asarray(rs.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32))
        c = da.empty((M, N), np.float32)
        lda, ldb = (M if ta else K), (K if tb else N)
        f = lambda: lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, lda, b._ptr, ldb, 0.0, c._ptr, N, _lib.F32)
        for _ in range(3):
            f()
        e0, e1 = _lib.Event(), _lib.Event()
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        ms = e0.elapsed_ms(e1) / 10
        print("%dx%dx%d %s: %8.1f us %6.1f TFLOP/s" % (M, N, K, name, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
