#!/usr/bin/env python3
"""Where the time goes inside dense_bwd0_allreduce_adam_kernel (data-parallel step, peer-to-peer transport, world 1).

Needs the debug library (`make -C tinynn-autograd_amd/csrc trace`, stamps compiled in with -DTNN_AR_TRACE):
    TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so TNN_FORCE_COMM=1 python tools/probes/ar_fused_trace.py
Prints, relative to the first workgroup's start (100 MHz wall clock, 10 ns steps): when the tile blocks started, finished
their product and had issued their sends; when the polling blocks started, finished stage A (rest of the arena), B, C."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402  (first: one HIP runtime in the process)
import tinynn_autograd_amd as tn  # noqa: E402
from tinynn_autograd_amd import _lib  # noqa: E402
from tinynn_autograd_amd.fused import MLPTrainer  # noqa: E402

os.environ["TNN_FORCE_COMM"] = "1"
comm = tn.dist.init_from_env()
widths, rows = [784, 256, 128, 10], 128
rng = np.random.default_rng(0)
x = tn.asarray(rng.standard_normal((rows, 784)).astype(np.float32))
y = tn.asarray(np.eye(10, dtype=np.float32)[rng.integers(0, 10, rows)])
t = MLPTrainer(widths, rows, loss="softmax_nll", optimizer="adam", lr=1e-3, comm=comm, force_dp=True)
lib = _lib.get()
fn = lib.cdll.tnn_debug_ar_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(4096, dtype=np.uint64)
fh = lib.cdll.tnn_debug_fh_trace
fh.argtypes = [ctypes.c_void_p, ctypes.c_int]
fbuf = np.zeros(1024, dtype=np.uint64)
acc, facc = [], []
for it in range(40):
    t.step(x, y)
    if it >= 20:
        assert fn(buf.ctypes.data, 4096) == 0
        acc.append(buf.reshape(1024, 4).astype(np.int64).copy())
        assert fh(fbuf.ctypes.data, 1024) == 0
        facc.append(fbuf.reshape(128, 8).astype(np.int64).copy())
# tile workgroups of the fused launch: 49 x 8 tiles of 16 x 32 (round 6; TNN_DW0_WIDE=0: 49 x 16 of 16 x 16), then P polling workgroups
n_dw, P = (392 if os.environ.get("TNN_DW0_WIDE", "1") != "0" else 784), 128
tr = np.stack(acc)                                     # [launch][block][stamp]
t0 = tr[:, :n_dw + P, 0].min(axis=1)[:, None, None]
rel = (tr - t0) / 100.0                                # us
tiles, poll = rel[:, :n_dw], rel[:, n_dw:n_dw + P]


def q(a):
    return "min %6.2f  med %6.2f  max %6.2f" % (np.min(a), np.median(a), np.median(np.max(a, axis=1)))


print("tile blocks   start        ", q(tiles[:, :, 0]))
print("tile blocks   product done ", q(tiles[:, :, 1]))
print("tile blocks   sends issued ", q(tiles[:, :, 2]))
print("polling blocks start       ", q(poll[:, :, 0]))
print("polling blocks stage A done", q(poll[:, :, 1]))
print("polling blocks stage B done", q(poll[:, :, 2]))
print("polling blocks stage C done", q(poll[:, :, 3]))

print("(launch: %d tile workgroups + %d polling workgroups; us relative to the launch's first workgroup entry; max = median over launches of the last workgroup)" % (n_dw, P))
# the hidden layer's forward + statistics launch (dense_fwd_head_kernel): 64 tile blocks, the last arrival does the tail —
# only in the forward-tail form of the step (TNN_DP_XCHG=0); the deferred form's forward is the single-GPU launch (no stamps)
ft = np.stack(facc)[:, :64]
if not ft.any():
    print("fwd1: the deferred-exchange step runs the plain forward launch (no statistics tail to stamp)")
    ft = None
if ft is not None:
    f0 = ft[:, :, 0].min(axis=1)[:, None]
    rel = lambda k: (ft[:, :, k] - f0) / 100.0
    print("fwd1 tile blocks start        ", q(rel(0)))
    print("fwd1 tile + partial logits    ", q(rel(1)))
    print("fwd1 partial logits acked     ", q(rel(2)))
    print("fwd1 ticket drawn             ", q(rel(3)))
    last = ft[:, :, 6] == 1
    lastrel = lambda k: np.array([(ft[i, last[i], k] - f0[i]) / 100.0 for i in range(ft.shape[0])]).ravel()
    print("fwd1 last block: ticket        med %6.2f" % np.median(lastrel(3)))
    print("fwd1 last block: statistics    med %6.2f" % np.median(lastrel(4)))
    print("fwd1 last block: exchange done med %6.2f" % np.median(lastrel(5)))

# which tile blocks are the slow ones?  (mean product-done time per block index over the traced launches)
import collections
mean_done = tiles[:, :, 1].mean(axis=0)
order = np.argsort(mean_done)
print("slowest tile blocks (index: us):", ", ".join("%d: %.2f" % (i, mean_done[i]) for i in order[-12:]))
print("fastest tile blocks (index: us):", ", ".join("%d: %.2f" % (i, mean_done[i]) for i in order[:6]))
byx = collections.defaultdict(list)
for i in range(n_dw):
    byx[i % 8].append(mean_done[i])
print("mean product-done time by block index mod 8 (XCD):", ", ".join("%d: %.2f" % (x, np.mean(byx[x])) for x in range(8)))
dur = (tiles[:, :, 1] - tiles[:, :, 0]).mean(axis=0)
print("product duration (start -> done) by position in the grid: first 256 workgroups %.2f, the rest %.2f" % (
    dur[:256].mean(), dur[256:].mean()))
