#!/usr/bin/env python3
"""Randomised check of every fp32 GEMM entry point against float64 numpy: all four layouts, ragged and aligned
shapes on both sides of the small-path / LDS-tiled boundary, bias+ReLU(sign mask) and mask epilogues, the fused
Dense backward and the first-layer backward with Adam folded in.  python tools/gemm_fuzz.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib

lib = _lib.get()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dims = [1, 3, 4, 7, 10, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128, 129, 200, 256, 300, 511, 512, 784, 1000, 1024]
worst = 0.0
for case in range(n_cases):
    M, N, K = (int(rs.choice(dims)) for _ in range(3))
    if rs.rand() < 0.15:
        M, N, K = int(rs.choice([512, 1024, 2048])), int(rs.choice([512, 1000, 2048])), int(rs.choice([256, 500, 1024]))
    ta, tb = int(rs.randint(2)), int(rs.randint(2))
    a = rs.randn(M, K).astype(np.float32); b = rs.randn(K, N).astype(np.float32)
    A = tn.asarray(np.ascontiguousarray(a.T) if ta else a); B = tn.asarray(np.ascontiguousarray(b.T) if tb else b)
    lda, ldb = (M if ta else K), (K if tb else N)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    bound = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64) + 1e-30
    kind = rs.randint(3)
    C = tn.empty((M, N))
    if kind == 0:
        lib.gemm(ta, tb, M, N, K, 1.0, A._ptr, lda, B._ptr, ldb, 0.0, C._ptr, N, _lib.F32)
        want = ref
    elif kind == 1:
        bias = rs.randn(N).astype(np.float32); Bi = tn.asarray(bias)
        lib.gemm_bias_act(ta, tb, M, N, K, A._ptr, lda, B._ptr, ldb, Bi._ptr, _lib.ACT_RELU, 1, C._ptr, N, _lib.F32)
        z = ref + bias
        want = np.where(z < 0, 0.0, np.abs(z))
        bound = bound + np.abs(bias)
    else:
        y = rs.randn(M, N).astype(np.float32); y[rs.rand(M, N) < 0.1] = -0.0
        Y = tn.asarray(y)
        lib.gemm_mask(ta, tb, M, N, K, A._ptr, lda, B._ptr, ldb, Y._ptr, N, C._ptr, N, _lib.F32)
        want = np.where(np.signbit(y), 0.0, ref)
    got = np.asarray(C, dtype=np.float64)
    if kind == 1:      # near-zero pre-activations may flip the ReLU: compare where the reference is clear of zero
        clear = np.abs(ref + bias) > 1e-5 * bound
        err = np.abs(got - want)[clear] / bound[clear] if clear.any() else np.zeros(1)
    else:
        err = np.abs(got - want) / bound
    worst = max(worst, float(err.max()))
    assert err.max() <= 3e-6, "case %d kind %d (%d,%d,%d) ta=%d tb=%d: rel err %g" % (case, kind, M, N, K, ta, tb, err.max())
# fused Dense backward and first-layer backward with Adam
for case in range(max(10, n_cases // 10)):
    rows, n_in, n_out = int(rs.choice([16, 80, 128, 256])), int(rs.choice([10, 64, 128, 784])), int(rs.choice([10, 16, 128, 256]))
    x = rs.randn(rows, n_in).astype(np.float32); x[rs.rand(rows, n_in) < 0.3] = -0.0
    dz = rs.randn(rows, n_out).astype(np.float32); w = rs.randn(n_in, n_out).astype(np.float32)
    X, DZ, W = tn.asarray(x), tn.asarray(dz), tn.asarray(w)
    dW, dB, dX = tn.empty((n_in, n_out)), tn.empty((n_out,)), tn.empty((rows, n_in))
    lib.dense_bwd(rows, n_in, n_out, X._ptr, DZ._ptr, W._ptr, dW._ptr, dB._ptr, dX._ptr, X._ptr, _lib.F32)
    np.testing.assert_allclose(np.asarray(dW), x.astype(np.float64).T @ dz, rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(np.asarray(dB), dz.astype(np.float64).sum(0), rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(np.asarray(dX), np.where(np.signbit(x), 0.0, dz.astype(np.float64) @ w.T), rtol=2e-5, atol=2e-4)
    # first layer + Adam: compare with the separate calls
    n_rest = int(rs.choice([0, 5, 1000]))
    P = tn.asarray(rs.randn(n_in * n_out + n_out + n_rest).astype(np.float32)); G = tn.asarray((rs.randn(P.size) * 1e-2).astype(np.float32))
    Mo, Vo = tn.asarray(np.abs(rs.randn(P.size)).astype(np.float32) * 1e-3), tn.asarray(np.abs(rs.randn(P.size)).astype(np.float32) * 1e-4)
    pows = tn.asarray(np.array([0.9 ** 3, 0.999 ** 3, 0, 0]), dtype=np.float64)
    P2, G2, M2, V2 = P.copy(), G.copy(), Mo.copy(), Vo.copy()
    nw = n_in * n_out
    off = lambda arr, o: arr._ptr + 4 * o
    lib.dense_bwd_first_adam(rows, n_in, n_out, X._ptr, DZ._ptr, off(G, 0), off(G, nw), off(P, 0), off(Mo, 0), off(Vo, 0),
                             off(P, nw), off(Mo, nw), off(Vo, nw), off(P, nw + n_out), off(G, nw + n_out), off(Mo, nw + n_out),
                             off(Vo, nw + n_out), n_rest, 1e-3, 0.9, 0.999, 1e-8, pows._ptr, _lib.F32)
    lib.gemm_tn_colsum(n_in, n_out, rows, X._ptr, n_in, DZ._ptr, n_out, off(G2, 0), n_out, off(G2, nw), _lib.F32)
    lib.adam_ex(P2._ptr, G2._ptr, M2._ptr, V2._ptr, P2.size, 1e-3, 0.9, 0.999, 1e-8, pows._ptr, None, _lib.F32, 0, None, None)
    for u, v, name in ((P, P2, "p"), (Mo, M2, "m"), (Vo, V2, "v"), (G, G2, "g")):
        np.testing.assert_allclose(np.asarray(u), np.asarray(v), rtol=1e-6, atol=1e-7, err_msg=name)
print("gemm_fuzz: %d GEMM cases ok, worst relative error %.2e" % (n_cases, worst))
