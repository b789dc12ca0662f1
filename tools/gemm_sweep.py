#!/usr/bin/env python3
"""Sweep the LDS-tiled MFMA GEMM's tile configurations (TNN_GEMM_CFG), split-K and the tile raster (TNN_GEMM_GROUP_M: M-tiles
per group; SWEEP_GROUP_M=1,2,4,8,16 — 1 gives every XCD's id range whole M panels) on the config-C shapes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd import device_array as da

SHAPES = [("NN fwd", 0, 0, 512, 4096, 4096), ("NT dX", 0, 1, 512, 4096, 4096), ("TN dW", 1, 0, 4096, 4096, 512)]
CFGS = {0: "128x128", 1: "64x128", 2: "64x64", 3: "128x64"}


def main():
    lib = _lib.get()
    rs = np.random.RandomState(0)
    for name, ta, tb, M, N, K in SHAPES:
        a = da.asarray(rs.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32))
        b = da.asarray(rs.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32))
        c = da.empty((M, N), np.float32)
        lda, ldb = (M if ta else K), (K if tb else N)
        for cfg, sk, gm in [(c_, s_, g_) for c_ in extra_cfgs() for s_ in [int(x) for x in os.environ.get("SWEEP_SPLITK", "1,2").split(",")]
                            for g_ in [int(x) for x in os.environ.get("SWEEP_GROUP_M", "8").split(",")]]:
            if True:
                os.environ["TNN_GEMM_CFG"], os.environ["TNN_GEMM_SPLITK"], os.environ["TNN_GEMM_GROUP_M"] = str(cfg), str(sk), str(gm)
                try:
                    for _ in range(2):
                        lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, lda, b._ptr, ldb, 0.0, c._ptr, N, _lib.F32)
                except Exception as e:
                    print("%-7s cfg %d split %d: %s" % (name, cfg, sk, e))
                    continue
                e0, e1 = _lib.Event(), _lib.Event()
                reps = 10
                e0.record()
                for _ in range(reps):
                    lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, lda, b._ptr, ldb, 0.0, c._ptr, N, _lib.F32)
                e1.record()
                ms = e0.elapsed_ms(e1) / reps
                print("%-7s cfg %d (%-8s) splitK %d group_m %2d : %8.1f us  %6.1f TFLOP/s" % (
                    name, cfg, CFGS.get(cfg, "?"), sk, gm, ms * 1e3, 2.0 * M * N * K / ms / 1e9))


def extra_cfgs():
    return [int(x) for x in os.environ.get("SWEEP_CFGS", "0,1,2,3").split(",")]


if __name__ == "__main__":
    main()
