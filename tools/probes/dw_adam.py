"""Timing probe: the bf16 dW GEMM with Adam in its epilogue (tnn_gemm_bf16_nt_adam) against the two launches it replaces,
8192 x 8192 x 512 (config E's weight-gradient shape).  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, bf16
import bench
lib = _lib.get()
M = N = int(os.environ.get("PROBE_N", "8192")); K = 512
rs = np.random.RandomState(0)
a = bf16.to_bf16(tn.asarray(rs.uniform(-1, 1, (M, K)).astype(np.float32)))
b = bf16.to_bf16(tn.asarray((rs.uniform(-1, 1, (N, K)) * 1e-2).astype(np.float32)))
P, M_, V_, G = tn.zeros((M, N)), tn.zeros((M, N)), tn.zeros((M, N)), tn.zeros((M, N))
W16, WT16 = tn.empty((M, N), np.uint16), tn.empty((N, M), np.uint16)
pows = tn.asarray(np.array([0.5, 0.5, 0, 0]), dtype=np.float64)
def gemm(): lib.gemm_bf16_nt(M, N, K, a._ptr, K, b._ptr, K, G._ptr, N, _lib.F32, None, 0, 0, None, 0)
def adam(): lib.adam_master_bf16_2d(P._ptr, G._ptr, M_._ptr, V_._ptr, W16._ptr, WT16._ptr, M, N, 1e-3, 0.9, 0.999, 1e-8, pows._ptr, 0)
def fused(keep):
    return lambda: lib.gemm_bf16_nt_adam(M, N, K, a._ptr, K, b._ptr, K, G._ptr if keep else None, P._ptr, M_._ptr, V_._ptr,
                                         W16._ptr, WT16._ptr, 1e-3, 0.9, 0.999, 1e-8, pows._ptr)
n = M * N
for name, fn, bytes_ in (("dW GEMM (fp32 out)", gemm, 4 * n), ("Adam + bf16 copies", adam, 32 * n),
                         ("fused, gradient not stored", fused(False), 28 * n), ("fused, gradient stored", fused(True), 32 * n)):
    us = bench.events_us(fn, 20)
    print("%-28s %8.1f us   %6.2f TB/s of its algorithmic bytes   %7.1f TFLOP/s" % (name, us, bytes_ / us / 1e6, 2.0 * M * N * K / us / 1e6))
def fused_no_wt():
    lib.gemm_bf16_nt_adam(M, N, K, a._ptr, K, b._ptr, K, None, P._ptr, M_._ptr, V_._ptr, W16._ptr, None, 1e-3, 0.9, 0.999, 1e-8, pows._ptr)
us = bench.events_us(fused_no_wt, 20)
print("%-28s %8.1f us   %6.2f TB/s" % ("fused, no transposed copy", us, 26 * n / us / 1e6))
