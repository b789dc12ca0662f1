"""The hidden layer's forward in front of the classifier head at 128 rows (256 -> 128 units, 10 classes): tile form
(tnn_dense_fwd_head_partials, 64 workgroups) against row-panel form (tnn_dense_fwd_rows_head_stats, 8 workgroups of 16 whole
rows).  Run under `rocprofv3 --kernel-trace --stats`: the kernels' average durations are the answer."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd import device_array as da
lib = _lib.get()
rs = np.random.RandomState(0)
M, K, N, C = 128, 256, 128, 10
a = da.asarray(rs.uniform(-1, 1, (M, K)).astype(np.float32))
w = da.asarray(rs.uniform(-0.1, 0.1, (K, N)).astype(np.float32))
b = da.asarray(rs.uniform(-0.1, 0.1, (N,)).astype(np.float32))
hw = da.asarray(rs.uniform(-0.1, 0.1, (N, C)).astype(np.float32))
hb = da.asarray(rs.uniform(-0.1, 0.1, (C,)).astype(np.float32))
out = da.empty((M, N), np.float32)
zpart = da.empty((8, M, C), np.float32)
zfull = da.empty((M, C), np.float32)
pairs = da.empty((64, 2), np.float32)
# a kernel between two calls that rewrites the input, so every call starts from memory as it does in a step
for it in range(300):
    lib.dense_fwd_head_partials(M, N, K, a._ptr, K, w._ptr, N, b._ptr, _lib.ACT_RELU, 1, out._ptr, N,
                                hw._ptr, C, zpart._ptr, _lib.F32)
    lib.dense_fwd_rows_head_stats(M, N, K, a._ptr, K, w._ptr, N, b._ptr, _lib.ACT_RELU, 1, out._ptr, N,
                                  hw._ptr, C, zfull._ptr, hb._ptr, pairs._ptr, _lib.F32)
_lib.synchronize()
z0 = np.asarray(zpart).sum(axis=0)
print("logits agree: max |tile - panel| = %.3g" % np.abs(z0 - np.asarray(zfull)).max())
