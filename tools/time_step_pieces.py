#!/usr/bin/env python3
"""Time each launch of the config-A training step in isolation: `reps` back-to-back replays inside one
hipGraph, HIP events on the library stream.  (GPU box only.)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib


def timed(fn, reps=200):
    for _ in range(3):
        fn()
    g = _lib.Graph()
    with g:
        for _ in range(reps):
            fn()
    g.launch()
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record(); g.launch(); e1.record()
    return e0.elapsed_ms(e1) / reps * 1e3


def main():
    lib = _lib.get()
    rs = np.random.RandomState(0)
    m, w = int(os.environ.get("ROWS", "128")), [784, 256, 128, 10]
    f = lambda *s: tn.asarray(rs.randn(*s).astype(np.float32))
    x, y = f(m, w[0]), tn.asarray(np.eye(10, dtype=np.float32)[rs.randint(0, 10, m)])
    W = [f(w[i], w[i + 1]) for i in range(3)]
    B = [f(w[i + 1]) for i in range(3)]
    act = [tn.empty((m, w[i + 1])) for i in range(3)]
    dact = [tn.empty((m, w[i + 1])) for i in range(3)]
    dW = [tn.empty((w[i], w[i + 1])) for i in range(3)]
    dB = [tn.empty((w[i + 1],)) for i in range(3)]
    stats, loss = tn.empty((2,)), tn.empty(())
    n = sum(a.size for a in W) + sum(b.size for b in B)
    p, g, mm, vv = (tn.zeros((n,)) for _ in range(4))
    pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
    F32 = _lib.F32
    pieces = {
        "fwd0 gemm_bias_act 128x256x784": lambda: lib.gemm_bias_act(0, 0, m, w[1], w[0], x._ptr, w[0], W[0]._ptr, w[1], B[0]._ptr, 1, 1, act[0]._ptr, w[1], F32),
        "fwd1 gemm_bias_act 128x128x256": lambda: lib.gemm_bias_act(0, 0, m, w[2], w[1], act[0]._ptr, w[1], W[1]._ptr, w[2], B[1]._ptr, 1, 1, act[1]._ptr, w[2], F32),
        "head (fwd2+nll+bwd2)": lambda: lib.mlp_head(m, w[2], w[3], act[1]._ptr, W[2]._ptr, B[2]._ptr, y._ptr, act[2]._ptr, dact[2]._ptr, stats._ptr, loss._ptr, dW[2]._ptr, dB[2]._ptr, dact[1]._ptr, F32),
        "dense_bwd layer1 (dW+db+dX)": lambda: lib.dense_bwd(m, w[1], w[2], act[0]._ptr, dact[1]._ptr, W[1]._ptr, dW[1]._ptr, dB[1]._ptr, dact[0]._ptr, act[0]._ptr, F32),
        "dense_bwd layer0 (dW+db)": lambda: lib.dense_bwd(m, w[0], w[1], x._ptr, dact[0]._ptr, W[0]._ptr, dW[0]._ptr, dB[0]._ptr, None, None, F32),
        "adam 235146": lambda: lib.adam(p._ptr, g._ptr, mm._ptr, vv._ptr, n, 1e-3, 0.9, 0.999, 1e-8, pows._ptr, None, F32),
        "nll_fused alone": lambda: lib.softmax_nll_fused(act[2]._ptr, y._ptr, m, 10, stats._ptr, loss._ptr, dact[2]._ptr, F32),
        "empty-ish: fill 16 floats": lambda: lib.fill(stats._ptr, 1.0, 2, F32),
        "fill 235146 floats (0.94 MB write)": lambda: lib.fill(p._ptr, 1.0, n, F32),
        "axpy 235146 (2 reads + 1 write)": lambda: lib.axpy(p._ptr, 0.5, g._ptr, n, F32),
        "sgd 235146 (2 reads + 1 write)": lambda: lib.sgd(p._ptr, g._ptr, n, 1e-3, F32),
        "unary exp 235146 (1 read + 1 write)": lambda: lib.ewise_unary(_lib.EXP, g._ptr, mm._ptr, n, F32),
    }
    total = 0.0
    for name, fn in pieces.items():
        us = timed(fn)
        if not name.startswith(("nll", "empty", "fill", "axpy", "sgd", "unary")):
            total += us
        print("%-36s %7.2f us" % (name, us))
    print("%-36s %7.2f us" % ("sum of the 6 step launches", total))


if __name__ == "__main__":
    main()
