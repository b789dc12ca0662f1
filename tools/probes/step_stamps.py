#!/usr/bin/env python3
"""Per-launch critical paths of the headline step (784-256-128-10, 128 rows, single GPU): in-kernel 100 MHz stamps of its four
launches from the debug library (`make -C tinynn-autograd_amd/csrc trace`, -DTNN_STEP_TRACE):
    TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so python3 tools/probes/step_stamps.py > profiles/r06_stepA_stamps.txt
For every launch, over its workgroups and 40 traced steps, relative to the launch's first workgroup entry (10 ns steps):
    entry | operands in registers (s_waitcnt vmcnt(0) behind the product's loads) | last MFMA + cross-wave sum read back | last store acknowledged
and how many workgroups the launch has against the 256 CUs.  (The stamps cost a few hundred ns per launch: the sum of the traced
launches is longer than the untraced step; the STAGES are what this is for.)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinynn_autograd_amd as tn  # noqa: E402
from tinynn_autograd_amd import _lib  # noqa: E402
from tinynn_autograd_amd.fused import MLPTrainer  # noqa: E402

widths, rows = [784, 256, 128, 10], int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(0)
x = tn.asarray((rng.random((rows, 784)) * (rng.random((rows, 784)) < 0.19)).astype(np.float32))
y = tn.asarray(np.eye(10, dtype=np.float32)[rng.integers(0, 10, rows)])
t = MLPTrainer(widths, rows, loss="softmax_nll", optimizer="adam", lr=1e-3)
t.keep_grads(False)
lib = _lib.get()
fg = lib.cdll.tnn_debug_step_trace
fg.argtypes = [ctypes.c_void_p, ctypes.c_int]
fh = lib.cdll.tnn_debug_step_trace_head
fh.argtypes = [ctypes.c_void_p, ctypes.c_int]
gbuf = np.zeros(4 * 1024 * 4, dtype=np.uint64)
hbuf = np.zeros(1024 * 4, dtype=np.uint64)
acc = []
for it in range(60):
    t.step(x, y)
    if it >= 20:
        assert fg(gbuf.ctypes.data, gbuf.size) == 0 and fh(hbuf.ctypes.data, hbuf.size) == 0
        g = gbuf.reshape(4, 1024, 4).astype(np.int64).copy()
        g[2] = hbuf.reshape(1024, 4).astype(np.int64)
        acc.append(g)
tr = np.stack(acc)                                     # [step][kernel][block][stamp]
names = ["fwd0  gemm_small_f32_kernel<.., 16, ..>  (784 -> 256, bias + ReLU)",
         "fwd1  gemm_small_f32_kernel<.., 4, ..>   (256 -> 128, bias + ReLU, partial logits)",
         "head  mlp_head_bwd_kernel<128, 10>       (loss, dz, dW2 / db2, dW1 / db1, dx1)",
         "bwd0  dense_bwd0_adam_kernel<4>          (dW0 / db0 + Adam over the arena)"]
grids = [8 * 16, 8 * 8, 16 + 16 * 8 + (8 * 8 if os.environ.get("TNN_HEAD_DX_WIDE", "1") != "0" else 8 * 16),
         49 * 8 if os.environ.get("TNN_DW0_WIDE", "1") != "0" else 49 * 16]
print("# rows %d; us relative to the launch's first workgroup entry; min / median / max over workgroups (median over 40 steps)" % rows)
if rows > 128:
    # the row-blocked head launch: 16 head + 128 dW1 workgroups, then ceil(rows / 16) x (8 or 16) dx tiles; only it is stamped
    dxc = 8 if (os.environ.get("TNN_HEAD_DX_WIDE", "1") != "0" and os.environ.get("TNN_HEAD_DX_WIDE_RB", "1") != "0") else 16
    grids[2] = 16 + 128 + (rows + 15) // 16 * dxc
for k in range(4):
    if rows > 128 and k != 2:
        continue
    n = min(grids[k], 1024)
    blk = tr[:, k, :n, :]
    live = blk[:, :, 0] > 0
    t0 = np.where(live, blk[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None, None]
    rel = (blk - t0) / 100.0
    print("%s   workgroups %d (%.2f per CU)" % (names[k], n, n / 256.0))
    for s, label in enumerate(("entry", "operands in registers", "last MFMA, sums read back", "last store acknowledged")):
        v = rel[:, :, s]
        ok = (blk[:, :, s] > 0)
        if not ok.any():
            continue
        vv = np.where(ok, v, np.nan)
        print("    %-28s min %6.2f   med %6.2f   max %6.2f" % (label, np.nanmedian(np.nanmin(vv, axis=1)), np.nanmedian(vv),
                                                                np.nanmedian(np.nanmax(vv, axis=1))))
    if k == 2:
        for role, lo, hi in (("16 head workgroups", 0, 16), ("128 dW1 tiles", 16, 144), ("%d dx1 tiles (those among the first 1024 workgroups)" % (n - 144), 144, n)):
            part = rel[:, lo:hi, 3]
            print("    (%-20s end: med %6.2f   max %6.2f; entry max %5.2f)" % (role, np.median(part), np.median(part.max(axis=1)),
                                                                              np.median(rel[:, lo:hi, 0].max(axis=1))))
        done = rel[:, :, 3].mean(axis=0)
        order = np.argsort(done)
        print("    slowest workgroups (index: end):", ", ".join("%d: %.2f" % (i, done[i]) for i in order[-10:]))
        dur = (rel[:, 144:n, 3] - rel[:, 144:n, 0])
        print("    dx tiles, entry -> end: min %.2f  10%% %.2f  med %.2f  90%% %.2f  max %.2f" % (
            np.median(dur.min(axis=1)), np.median(np.percentile(dur, 10, axis=1)), np.median(dur), np.median(np.percentile(dur, 90, axis=1)),
            np.median(dur.max(axis=1))))
        byx = [np.median(dur[:, x::8]) for x in range(8)]
        print("    dx tiles, median duration by XCD (index mod 8):", " ".join("%.2f" % v for v in byx))
    if k == 3:
        done = rel[:, :, 3].mean(axis=0)
        order = np.argsort(done)
        print("    slowest workgroups (index: end):", ", ".join("%d: %.2f" % (i, done[i]) for i in order[-8:]))
        if n == 784:
            print("    end by position in the grid: first 256 %.2f, second 256 %.2f, third 256 %.2f, last 16 %.2f" % (
                done[:256].mean(), done[256:512].mean(), done[512:768].mean(), done[768:].mean()))
        else:
            print("    end by position in the grid: first 256 %.2f, the other %d %.2f" % (done[:256].mean(), n - 256, done[256:].mean()))
