#!/usr/bin/env python3
"""Minimal workload for rocprofv3 --pmc passes: the three config-C GEMM shapes, a few launches each."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd import device_array as da

lib = _lib.get()
rs = np.random.RandomState(0)
for ta, tb, M, N, K in ((0, 0, 512, 4096, 4096), (0, 1, 512, 4096, 4096), (1, 0, 4096, 4096, 512)):
    a = da.asarray(rs.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32))
    b = da.asarray(rs.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32))
    c = da.empty((M, N), np.float32)
    for _ in range(int(os.environ.get("REPS", "4"))):
        lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, (M if ta else K), b._ptr, (K if tb else N), 0.0, c._ptr, N, _lib.F32)
_lib.synchronize()
