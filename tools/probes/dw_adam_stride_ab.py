"""Does the power-of-two row stride of the 8192-wide fp32 master arrays cost bandwidth?  The bf16 dW + Adam launch (K = 512) on
[rows x cols] = 8192 x 8192, 8192 x 8320 (row stride 33,280 B instead of 32,768) and 8320 x 8192 (the same stride as the square case,
more rows), alternating segments in ONE process, HIP events; three parameter sets per shape in rotation.  Reported: us per launch and
ns per 1000 parameters (the shapes differ by 1.6 % in size)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, bf16

lib = _lib.get()
K = 512
REPS, ROUNDS = int(os.environ.get("AB_REPS", "9")), int(os.environ.get("AB_ROUNDS", "8"))
rs = np.random.RandomState(3)
ev0, ev1 = _lib.Event(), _lib.Event()
pows = tn.asarray(np.array([0.9, 0.999, 0.0, 0.0]), dtype=np.float64)
SHAPES = [(8192, 8192), (8192, 8320), (8320, 8192), (8320, 8320)]
data = {}
for (R, C) in SHAPES:
    sets = [dict(p=tn.asarray(rs.uniform(-0.03, 0.03, (R, C)).astype(np.float32)), m=tn.zeros((R, C), np.float32), v=tn.zeros((R, C), np.float32),
                 w=tn.empty((R, C), np.uint16), wt=tn.empty((C, R), np.uint16)) for _ in range(3)]
    inT = bf16.to_bf16(rs.uniform(0, 1, (R, K)).astype(np.float32))
    dzT = bf16.to_bf16(rs.uniform(-1e-3, 1e-3, (C, K)).astype(np.float32))
    data[(R, C)] = (sets, inT, dzT)


def timed(shape, copies):
    R, C = shape
    sets, inT, dzT = data[shape]

    def call(i):
        s = sets[i % 3]
        lib.gemm_bf16_nt_adam(R, C, K, inT._ptr, K, dzT._ptr, K, None, s["p"]._ptr, s["m"]._ptr, s["v"]._ptr,
                              s["w"]._ptr if "w" in copies else None, s["wt"]._ptr if "t" in copies else None, 1e-3, 0.9, 0.999, 1e-8, pows._ptr)
    for i in range(3):
        call(i)
    ev0.record()
    for i in range(REPS):
        call(i)
    ev1.record()
    return ev0.elapsed_ms(ev1) / REPS * 1e3


for copies in ("wt", "t"):
    res = {s: [] for s in SHAPES}
    for r in range(ROUNDS):
        for s in (SHAPES if r % 2 == 0 else SHAPES[::-1]):
            res[s].append(timed(s, copies))
    print("bf16 copies written: %s" % {"wt": "W and W^T (28 B/param)", "t": "W^T only (26 B/param)"}[copies])
    for s in SHAPES:
        v = np.array(res[s])
        bpp = 24.0 + 2 * len(copies)
        print("  %5d x %5d   median %7.1f us  min %7.1f  max %7.1f   %6.2f ns per 1000 params   %.2f TB/s"
              % (s[0], s[1], np.median(v), v.min(), v.max(), np.median(v) * 1e6 / (s[0] * s[1]), bpp * s[0] * s[1] / np.median(v) / 1e6))
