"""bf16 storage helpers for the bf16 path (BASELINE.json configs[4]).  numpy has no bfloat16, so bf16 buffers are
DeviceArrays of dtype uint16 holding the raw bit patterns; they support no arithmetic — only the entry points
below (cast, transpose, the K-contiguous GEMM) and the bf16 trainer consume them."""

import numpy as np

from . import _lib
from . import device_array as da


def to_bf16(x):
    """float32 DeviceArray / ndarray -> bf16 (round to nearest even) DeviceArray (dtype uint16)."""
    x = da.asarray(x, dtype=np.float32)._contig()
    out = da.empty(x.shape, np.uint16)
    _lib.get().cast_bf16(x._ptr, out._ptr, x.size, 1)
    return out


def to_f32(x16):
    out = da.empty(x16.shape, np.float32)
    _lib.get().cast_bf16(x16._ptr, out._ptr, x16.size, 0)
    return out


def transpose(x16):
    r, c = x16.shape
    out = da.empty((c, r), np.uint16)
    _lib.get().transpose_bf16(x16._ptr, out._ptr, r, c)
    return out


def gemm_nt(a16, b16, out_dtype=np.float32, bias=None, relu=False, relu_sign=False, mask=None):
    """C[M,N] = A[M,K] @ B[N,K]^T (bf16 in, fp32 accumulate); out_dtype float32 or 'bf16' (uint16 storage)."""
    M, K = a16.shape
    N, K2 = b16.shape
    if K != K2:
        raise ValueError("gemm_nt: K mismatch %d vs %d" % (K, K2))
    bf_out = np.dtype(out_dtype) == np.uint16
    out = da.empty((M, N), np.uint16 if bf_out else np.float32)
    _lib.get().gemm_bf16_nt(M, N, K, a16._ptr, K, b16._ptr, K, out._ptr, N, _lib.BF16 if bf_out else _lib.F32,
                            None if bias is None else bias._ptr, _lib.ACT_RELU if relu else _lib.ACT_NONE,
                            int(relu_sign), None if mask is None else mask._ptr, N)
    return out


def gemm_nt_t(a16, b16, bias=None, relu=False, relu_sign=False, mask=None):
    """gemm_nt with bf16 output that also returns C^T [N, M], written by the producing epilogue (tnn_gemm_bf16_nt_t):
    (C, C_t)."""
    M, K = a16.shape
    N, K2 = b16.shape
    if K != K2:
        raise ValueError("gemm_nt_t: K mismatch %d vs %d" % (K, K2))
    out, out_t = da.empty((M, N), np.uint16), da.empty((N, M), np.uint16)
    _lib.get().gemm_bf16_nt_t(M, N, K, a16._ptr, K, b16._ptr, K, out._ptr, N, None if bias is None else bias._ptr,
                              _lib.ACT_RELU if relu else _lib.ACT_NONE, int(relu_sign), None if mask is None else mask._ptr, N,
                              out_t._ptr, M)
    return out, out_t


def round_to_bf16(x):
    """Host emulation of the device's f32 -> bf16 -> f32 round trip (for oracles in tests)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32)
