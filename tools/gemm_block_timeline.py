#!/usr/bin/env python3
"""Per-workgroup timeline of the LDS-tiled fp32 GEMM (config-C shapes) from the kernel's own timestamps.

Needs the debug library:  make -C tinynn-autograd_amd/csrc trace   (libtnn_hip_trace.so: tnn_gemm.hip compiled with
-DTNN_GEMM_TRACE; thread 0 of every workgroup writes entry / K-loop start / K-loop end / stores-acknowledged times from
the 100 MHz clock plus HW_ID and XCC_ID).  Prints, per shape: how the workgroups spread over the CUs, the time spent
in prologue, K loop and epilogue, and one CU's sequence of workgroups — which is how the lock-step of equal-length
workgroups (and what it costs the 4096x4096x512 dW GEMM) shows up.  Output of round 1: profiles/r01_gemm_block_timeline.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinynn_autograd_amd import _lib

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libtnn_hip_trace.so")
if not os.path.exists(_lib.LIB_PATH):
    sys.exit("build the debug library first: make -C tinynn-autograd_amd/csrc trace")
from tinynn_autograd_amd import device_array as da

SHAPES = (("NN fwd", 0, 0, 512, 4096, 4096), ("NT dX", 0, 1, 512, 4096, 4096), ("TN dW", 1, 0, 4096, 4096, 512))


def main():
    lib = _lib.get()
    rs = np.random.RandomState(0)
    for name, ta, tb, M, N, K in SHAPES:
        a = da.asarray(rs.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32))
        b = da.asarray(rs.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32))
        c = da.empty((M, N), np.float32)
        nb = ((M + 127) // 128) * ((N + 63) // 64)          # the 128x64 configuration's grid
        tr = da.asarray(np.zeros((nb, 8), np.int64))
        os.environ["TNN_GEMM_TRACE_PTR"] = str(tr._ptr)
        os.environ["TNN_GEMM_CFG"], os.environ["TNN_GEMM_SPLITK"] = "3", "1"
        for _ in range(3):                                   # the last launch's records survive
            lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, (M if ta else K), b._ptr, (K if tb else N), 0.0, c._ptr, N, _lib.F32)
        _lib.synchronize()
        t = tr.numpy().astype(np.int64)
        t0 = t[:, 0].min()
        st, lp, le, en = [(t[:, i] - t0) / 100.0 for i in range(4)]      # us
        hw, xcc = t[:, 4] & 0xffffffff, (t[:, 4] >> 32) & 0xf
        cu = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 10 + ((hw >> 8) & 0xf)   # XCC, SE, SH, CU
        per_cu = {}
        for i in range(nb):
            per_cu.setdefault(int(cu[i]), []).append(i)
        cnt = np.array([len(v) for v in per_cu.values()])
        ideal = 2.0 * M * N * K / 157.3e6
        print("== %s  %dx%dx%d: %d workgroups on %d CUs (%d..%d each), span %.1f us (%.1f us at the 157.3 TF peak)"
              % (name, M, N, K, nb, len(per_cu), cnt.min(), cnt.max(), en.max(), ideal))
        for label, d in (("prologue", lp - st), ("K loop", le - lp), ("epilogue", en - le), ("workgroup", en - st)):
            print("   %-10s mean %6.2f  p50 %6.2f  min %6.2f  max %6.2f us" % (label, d.mean(), np.median(d), d.min(), d.max()))
        print("   starts: p50 %.1f  max %.1f us" % (np.median(st), st.max()))
        one = sorted(per_cu.items())[0][1]
        print("   one CU (entry, loop start, loop end, stores acknowledged):")
        for i in sorted(one, key=lambda i: st[i]):
            print("     wg %4d  %7.2f %7.2f %7.2f %7.2f" % (i, st[i], lp[i], le[i], en[i]))


if __name__ == "__main__":
    main()
