import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd import device_array as da
lib = _lib.get()
rs = np.random.RandomState(0)
M, N, K = 512, 4096, 4096
lim = float(np.sqrt(6.0 / (K + N)))
sets = {"a U(-1,1), b U(-1,1)": (rs.uniform(-1, 1, (M, K)), rs.uniform(-1, 1, (K, N))),
        "a U(0,1),  b U(-1,1)": (rs.rand(M, K), rs.uniform(-1, 1, (K, N))),
        "a U(-1,1), b Xavier": (rs.uniform(-1, 1, (M, K)), rs.uniform(-lim, lim, (K, N))),
        "a U(0,1),  b Xavier (bench)": (rs.rand(M, K), rs.uniform(-lim, lim, (K, N))),
        "zeros": (np.zeros((M, K)), np.zeros((K, N)))}
ops = {k: (da.asarray(a.astype(np.float32)), da.asarray(b.astype(np.float32))) for k, (a, b) in sets.items()}
c = da.empty((M, N), np.float32)
res = {k: [] for k in ops}
for rnd in range(8):
    for k, (a, b) in ops.items():
        e0, e1 = _lib.Event(), _lib.Event()
        e0.record()
        for _ in range(10):
            lib.gemm(0, 0, M, N, K, 1.0, a._ptr, K, b._ptr, N, 0.0, c._ptr, N, _lib.F32)
        e1.record()
        if rnd: res[k].append(e0.elapsed_ms(e1) / 10 * 1e3)
for k, v in res.items():
    v = sorted(v); med = v[len(v)//2]
    print("%-30s median %7.1f us  %6.1f TFLOP/s" % (k, med, 2.0*M*N*K/med/1e6))
# ---- the same product eager (back-to-back launches) against replayed from a hipGraph of 20 launches
a, b = ops["a U(0,1),  b Xavier (bench)"]
def call():
    lib.gemm(0, 0, M, N, K, 1.0, a._ptr, K, b._ptr, N, 0.0, c._ptr, N, _lib.F32)
g = _lib.Graph()
with g:
    for _ in range(20):
        call()
g.launch()
out = {"eager": [], "graph": []}
for rnd in range(8):
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    if rnd: out["eager"].append(e0.elapsed_ms(e1) / 20 * 1e3)
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    g.launch()
    e1.record()
    if rnd: out["graph"].append(e0.elapsed_ms(e1) / 20 * 1e3)
for k, v in out.items():
    v = sorted(v); med = v[len(v)//2]
    print("%-30s median %7.1f us  %6.1f TFLOP/s  (min %.1f max %.1f)" % (k + " x20", med, 2.0*M*N*K/med/1e6, v[0], v[-1]))
