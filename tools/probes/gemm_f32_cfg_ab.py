"""fp32 tiled GEMM on config C's shapes, alternating A/B inside ONE process (segments of REPS calls, HIP events; a plain
sweep drifts by 2-4 % along its own order on this pool): tile 64 x 128 (cfg 1) against 128 x 64 (cfg 3) and the raster group
(TNN_GEMM_GROUP_M 8 against 16 / 32), per layout.  Operands: activations uniform [0, 1), weights +-0.027 (the step's)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from tinynn_autograd_amd import _lib
from tinynn_autograd_amd import device_array as da

lib = _lib.get()
rs = np.random.RandomState(0)
REPS, ROUNDS = 10, 10
ev0, ev1 = _lib.Event(), _lib.Event()
SHAPES = [("NN fwd", 0, 0, 512, 4096, 4096), ("NT dX", 0, 1, 512, 4096, 4096), ("TN dW", 1, 0, 4096, 4096, 512)]
FORMS = [("cfg 1 (64 x 128), group 8", "1", "8"), ("cfg 3 (128 x 64), group 8", "3", "8"), ("cfg 3 (128 x 64), group 16", "3", "16"),
         ("cfg 3 (128 x 64), group 32", "3", "32")]

for name, ta, tb, M, N, K in SHAPES:
    a_shape, b_shape = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
    a = da.asarray(rs.uniform(0, 1, a_shape).astype(np.float32))
    bs = [da.asarray(rs.uniform(-0.027, 0.027, b_shape).astype(np.float32)) for _ in range(2)]
    c = da.empty((M, N), np.float32)
    lda, ldb = (M if ta else K), (K if tb else N)

    def seg(cfg, gm):
        os.environ["TNN_GEMM_CFG"], os.environ["TNN_GEMM_GROUP_M"] = cfg, gm
        for i in range(2):
            lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, lda, bs[i % 2]._ptr, ldb, 0.0, c._ptr, N, _lib.F32)
        ev0.record()
        for i in range(REPS):
            lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, lda, bs[i % 2]._ptr, ldb, 0.0, c._ptr, N, _lib.F32)
        ev1.record()
        return ev0.elapsed_ms(ev1) / REPS * 1e3

    res = {f[0]: [] for f in FORMS}
    for r in range(ROUNDS):
        order = FORMS if r % 2 == 0 else FORMS[::-1]
        for label, cfg, gm in order:
            res[label].append(seg(cfg, gm))
    print("%s  %d x %d x %d" % (name, M, N, K))
    for label, _, _ in FORMS:
        v = np.array(res[label])
        print("  %-28s median %7.1f us  min %7.1f  max %7.1f   %6.1f TFLOP/s" % (label, np.median(v), v.min(), v.max(),
                                                                                 2.0 * M * N * K / np.median(v) / 1e6))
os.environ.pop("TNN_GEMM_CFG", None)
os.environ.pop("TNN_GEMM_GROUP_M", None)
