"""configs[4] kernels, A/B inside ONE process (alternating segments, HIP events — see e_step_ab.py for why):
  * the 512 x 8192 x 8192 bf16 product with and without the transposed second output (tnn_gemm_bf16_nt_t vs _nt) for the plain,
    bias + ReLU and mask epilogues, against a separate tnn_transpose_bf16 launch;
  * the 8192 x 8192 x 512 dW + Adam launch alone and followed by the bias launch; four bias launches against one
    (tnn_bias_bf16_adam_multi);
  * the prep launch (tnn_mse_bf16_prep) against tnn_mse_bf16_tick + two transposes.
Weights rotate over three matrices (402 MB > the memory-side cache) like in the training step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, bf16

lib = _lib.get()
M, W = 512, 8192
REPS, ROUNDS = 12, 8
rs = np.random.RandomState(3)
ev0, ev1 = _lib.Event(), _lib.Event()


def rnd16(shape, lo=-1.0, hi=1.0):
    return bf16.to_bf16(rs.uniform(lo, hi, shape).astype(np.float32))


a = rnd16((M, W), 0.0, 1.0)
Bs = [rnd16((W, W), -0.03, 0.03) for _ in range(3)]
bias = tn.asarray(rs.randn(W).astype(np.float32))
c, ct = tn.empty((M, W), np.uint16), tn.empty((W, M), np.uint16)
act = bf16.gemm_nt(a, Bs[0], out_dtype=np.uint16, bias=bias, relu=True, relu_sign=True)


def timed(fn):
    for i in range(2):
        fn(i)
    ev0.record()
    for i in range(REPS):
        fn(i)
    ev1.record()
    return ev0.elapsed_ms(ev1) / REPS * 1e3


def ab(title, forms):
    res = {k: [] for k in forms}
    for r in range(ROUNDS):
        keys = list(forms) if r % 2 == 0 else list(forms)[::-1]
        for k in keys:
            res[k].append(timed(forms[k]))
    print(title)
    for k in forms:
        v = np.array(res[k])
        print("  %-46s median %7.1f us  min %7.1f  max %7.1f" % (k, np.median(v), v.min(), v.max()))


def gemm(epi, with_t, separate_t=False):
    def fn(i):
        B = Bs[i % 3]
        b_, act_, rs_ = (bias._ptr, _lib.ACT_RELU, 1) if epi == "relu" else (None, _lib.ACT_NONE, 0)
        y_ = act._ptr if epi == "mask" else None
        if with_t:
            lib.gemm_bf16_nt_t(M, W, W, a._ptr, W, B._ptr, W, c._ptr, W, b_, act_, rs_, y_, W, ct._ptr, M)
        else:
            lib.gemm_bf16_nt(M, W, W, a._ptr, W, B._ptr, W, c._ptr, W, _lib.BF16, b_, act_, rs_, y_, W)
            if separate_t:
                lib.transpose_bf16(c._ptr, ct._ptr, M, W)
    return fn


for epi in ("plain", "relu", "mask"):
    ab("512 x 8192 x 8192 -> bf16, epilogue %s" % epi,
       {"C only": gemm(epi, False), "C and C^T from one launch": gemm(epi, True),
        "C, then a transpose launch": gemm(epi, False, True)})

# dW + Adam (+ bias)
inT, dzT = rnd16((W, M), 0.0, 1.0), rnd16((W, M), -1e-3, 1e-3)
dz = bf16.transpose(dzT)
P = [tn.asarray(rs.uniform(-0.03, 0.03, (W, W)).astype(np.float32)) for _ in range(2)]
Mo = [tn.zeros((W, W), np.float32) for _ in range(2)]
Vo = [tn.zeros((W, W), np.float32) for _ in range(2)]
W16 = [tn.empty((W, W), np.uint16) for _ in range(2)]
WT16 = [tn.empty((W, W), np.uint16) for _ in range(2)]
bp, bm, bv = tn.asarray(rs.randn(W).astype(np.float32)), tn.zeros((W,), np.float32), tn.zeros((W,), np.float32)
bw16, db = tn.empty((W,), np.uint16), tn.empty((W,), np.float32)
pows = tn.asarray(np.array([0.9, 0.999, 0.0, 0.0]), dtype=np.float64)


def dw_only(i):
    k = i % 2
    lib.gemm_bf16_nt_adam(W, W, M, inT._ptr, M, dzT._ptr, M, None, P[k]._ptr, Mo[k]._ptr, Vo[k]._ptr, W16[k]._ptr, WT16[k]._ptr,
                          1e-3, 0.9, 0.999, 1e-8, pows._ptr)


def dw_then_bias(i):
    dw_only(i)
    lib.bias_bf16_adam(dz._ptr, M, W, db._ptr, bp._ptr, bm._ptr, bv._ptr, bw16._ptr, 1e-3, 0.9, 0.999, 1e-8, pows._ptr)


ab("8192 x 8192 x 512 dW + Adam", {"dW + Adam alone": dw_only, "dW + Adam, then the bias launch": dw_then_bias})

import ctypes                                                                     # noqa: E402
arr4 = lambda a: (ctypes.c_void_p * 4)(*[a._ptr] * 4)                            # noqa: E731


def bias_four(i):
    for _ in range(4):
        lib.bias_bf16_adam(dz._ptr, M, W, db._ptr, bp._ptr, bm._ptr, bv._ptr, bw16._ptr, 1e-3, 0.9, 0.999, 1e-8, pows._ptr)


def bias_multi(i):
    lib.bias_bf16_adam_multi(4, arr4(dz), M, (ctypes.c_int64 * 4)(W, W, W, W), arr4(db), arr4(bp), arr4(bm), arr4(bv), arr4(bw16),
                             1e-3, 0.9, 0.999, 1e-8, pows._ptr)


ab("bias gradient + Adam of four 8192-wide layers", {"four launches": bias_four, "one launch": bias_multi})

# prep
pred, y, x = rnd16((M, W)), rnd16((M, W)), rnd16((M, W), 0.0, 1.0)
dzo, dzt, xt = tn.empty((M, W), np.uint16), tn.empty((W, M), np.uint16), tn.empty((W, M), np.uint16)
loss = tn.empty((2,), np.float32)
ws, ticket = tn.empty((M // 64 * (W // 64),), np.float64), tn.asarray(np.zeros(32, np.int64))


def prep_new(i):
    lib.mse_bf16_prep(pred._ptr, y._ptr, M, W, M, loss._ptr, loss._ptr + 4, dzo._ptr, dzt._ptr, x._ptr, W, xt._ptr, ws._ptr,
                      ticket._ptr, None, 0.9, 0.999)


def prep_old(i):
    lib.mse_bf16_tick(pred._ptr, y._ptr, M * W, M, loss._ptr, loss._ptr + 4, dzo._ptr, None, 0.9, 0.999)
    lib.transpose_bf16(dzo._ptr, dzt._ptr, M, W)
    lib.transpose_bf16(x._ptr, xt._ptr, M, W)


ab("loss + dz + dz^T + x^T (512 x 8192)", {"mse + partial sums + two transposes (4 launches)": prep_old,
                                            "prep (1 launch)": prep_new})
