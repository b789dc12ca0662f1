"""Fixed-overhead probe: the fp32 TN product 4096 x 4096 x K for several K — time = a + b K; a is what the 2048 tiles'
prologues / epilogues / dispatch cost, b K the steady loop.  Also M x 4096 x 512 for several M (tiles scale, K fixed)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd import device_array as da
lib = _lib.get()
rs = np.random.RandomState(0)


def run(M, N, K, ta=1, tb=0):
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
    return e0.elapsed_ms(e1) / 10 * 1e3


for K in (128, 256, 512, 1024, 2048, 4096):
    us = run(4096, 4096, K)
    print("TN 4096x4096x%-5d %8.1f us %6.1f TFLOP/s  (%.1f us per 32-deep K-tile step)" % (K, us, 2.0 * 4096 * 4096 * K / us / 1e6, us / (K / 32)), flush=True)
for M in (512, 1024, 2048, 4096, 8192):
    us = run(M, 4096, 512)
    print("TN %5dx4096x512  %8.1f us %6.1f TFLOP/s  (%d tiles)" % (M, us, 2.0 * M * 4096 * 512 / us / 1e6, (M // 128) * 64), flush=True)
