#!/usr/bin/env python3
"""Time the bf16 K-contiguous GEMM on the config-E shapes (8192-wide layers, 512 rows)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinynn_autograd_amd import _lib, bf16
from tinynn_autograd_amd import device_array as da

rs = np.random.RandomState(0)
for name, M, N, K, out in (("fwd/dX 512x8192x8192 -> bf16", 512, 8192, 8192, np.uint16),
                           ("dW 8192x8192x512 -> f32", 8192, 8192, 512, np.float32),
                           ("square 4096^3 -> f32", 4096, 4096, 4096, np.float32),
                           ("square 8192^3 -> bf16", 8192, 8192, 8192, np.uint16)):
    A = bf16.to_bf16(rs.uniform(-1, 1, (M, K)).astype(np.float32))
    B = bf16.to_bf16(rs.uniform(-1, 1, (N, K)).astype(np.float32))
    for _ in range(2):
        c = bf16.gemm_nt(A, B, out_dtype=out)
    e0, e1 = _lib.Event(), _lib.Event()
    reps = 10
    e0.record()
    for _ in range(reps):
        c = bf16.gemm_nt(A, B, out_dtype=out)
    e1.record()
    ms = e0.elapsed_ms(e1) / reps
    print("%-32s %9.1f us  %7.1f TFLOP/s" % (name, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
