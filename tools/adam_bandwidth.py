import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib
lib = _lib.get()
n = 8192 * 8192
P, G, M, V = (tn.zeros((n,)) for _ in range(4))
G[...] = 0.01
W16 = tn.empty((n,), np.uint16); WT = tn.empty((n,), np.uint16)
pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
def t(fn, reps=5):
    fn(); _lib.synchronize()
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record()
    return e0.elapsed_ms(e1) / reps * 1e3
print("fp32 adam        %8.1f us  (28 B/param -> %.2f TB/s)" % ((a := t(lambda: lib.adam(P._ptr, G._ptr, M._ptr, V._ptr, n, 1e-3, .9, .999, 1e-8, pows._ptr, None, _lib.F32))), n * 28 / a / 1e6))
print("bf16 flat        %8.1f us  (30 B/param -> %.2f TB/s)" % ((a := t(lambda: lib.adam_master_bf16(P._ptr, G._ptr, M._ptr, V._ptr, W16._ptr, n, 1e-3, .9, .999, 1e-8, pows._ptr))), n * 30 / a / 1e6))
print("bf16 tiled +WT   %8.1f us  (32 B/param -> %.2f TB/s)" % ((a := t(lambda: lib.adam_master_bf16_2d(P._ptr, G._ptr, M._ptr, V._ptr, W16._ptr, WT._ptr, 8192, 8192, 1e-3, .9, .999, 1e-8, pows._ptr, 1))), n * 32 / a / 1e6))
print("bf16 tiled noWT  %8.1f us  (30 B/param -> %.2f TB/s)" % ((a := t(lambda: lib.adam_master_bf16_2d(P._ptr, G._ptr, M._ptr, V._ptr, W16._ptr, None, 8192, 8192, 1e-3, .9, .999, 1e-8, pows._ptr, 1))), n * 30 / a / 1e6))
print("transpose bf16   %8.1f us  (4 B/elem -> %.2f TB/s)" % ((a := t(lambda: lib.transpose_bf16(W16._ptr, WT._ptr, 8192, 8192))), n * 4 / a / 1e6))
