"""Can a 512-row bf16 GEMM hide under the HBM-bound dW + Adam launch?  Feasibility probe for overlapping the NEXT step's forward
products with this step's dW launches (configs[4]): the GEMM goes to the communication stream (tnn_comm_chain_begin / _end, a
one-rank RCCL communicator only provides that stream), the dW + Adam launch to the library stream, HIP events around both.
  dma : the 128 x 128 LDS-DMA kernel (fp32 output: the shape the split-K kernel does not take) — 64 KB of LDS, no workgroup
        waits for another
  sk  : the 256 x 128 split-K kernel (bf16 output) — 160 KB of LDS and partner workgroups that must be co-resident: run LAST and
        only with OVERLAP_SK=1 (a hand-off that times out poisons the process's fault word)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime per process)

import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, bf16
from tinynn_autograd_amd.dist import RcclCommunicator

lib = _lib.get()
comm = RcclCommunicator(0, 1, RcclCommunicator.new_unique_id())
M, W = 512, 8192
rs = np.random.RandomState(3)
r16 = lambda shape, lo, hi: bf16.to_bf16(rs.uniform(lo, hi, shape).astype(np.float32))     # noqa: E731
a, Bs = r16((M, W), 0.0, 1.0), [r16((W, W), -0.03, 0.03) for _ in range(3)]
c32, c16 = tn.empty((M, W), np.float32), tn.empty((M, W), np.uint16)
inT, dzT = r16((W, M), 0.0, 1.0), r16((W, M), -1e-3, 1e-3)
P = [tn.asarray(rs.uniform(-0.03, 0.03, (W, W)).astype(np.float32)) for _ in range(2)]
Mo, Vo = [tn.zeros((W, W), np.float32) for _ in range(2)], [tn.zeros((W, W), np.float32) for _ in range(2)]
W16, WT16 = [tn.empty((W, W), np.uint16) for _ in range(2)], [tn.empty((W, W), np.uint16) for _ in range(2)]
pows = tn.asarray(np.array([0.9, 0.999, 0.0, 0.0]), dtype=np.float64)
ev0, ev1 = _lib.Event(), _lib.Event()


def dw(i):
    k = i % 2
    lib.gemm_bf16_nt_adam(W, W, M, inT._ptr, M, dzT._ptr, M, None, P[k]._ptr, Mo[k]._ptr, Vo[k]._ptr, W16[k]._ptr, WT16[k]._ptr,
                          1e-3, 0.9, 0.999, 1e-8, pows._ptr)


def gemm(i, sk):
    B = Bs[i % 3]
    if sk:
        lib.gemm_bf16_nt(M, W, W, a._ptr, W, B._ptr, W, c16._ptr, W, _lib.BF16, None, _lib.ACT_NONE, 0, None, W)
    else:
        lib.gemm_bf16_nt(M, W, W, a._ptr, W, B._ptr, W, c32._ptr, W, _lib.F32, None, _lib.ACT_NONE, 0, None, W)


def timed(fn, reps=10):
    for i in range(2):
        fn(i)
    _lib.synchronize()
    ev0.record()
    for i in range(reps):
        fn(i)
    ev1.record()
    return ev0.elapsed_ms(ev1) / reps * 1e3


def both(sk, n_gemm):
    def fn(i):
        lib.comm_chain_begin()
        for j in range(n_gemm):
            gemm(i + j, sk)
        lib.comm_chain_end()
        dw(i)
        lib.comm_join()
    return fn


def serial(sk, n_gemm):
    def fn(i):
        for j in range(n_gemm):
            gemm(i + j, sk)
        dw(i)
    return fn


for sk in ([False] + ([True] if os.environ.get("OVERLAP_SK") == "1" else [])):
    name = "split-K 256 x 128 (bf16 out)" if sk else "LDS-DMA 128 x 128 (fp32 out)"
    print("GEMM kernel: %s" % name)
    print("  dW + Adam alone                       %7.1f us" % np.median([timed(dw) for _ in range(5)]))
    print("  GEMM alone                            %7.1f us" % np.median([timed(lambda i: gemm(i, sk)) for _ in range(5)]))
    for n in (1, 3):
        s = np.median([timed(serial(sk, n)) for _ in range(5)])
        o = np.median([timed(both(sk, n)) for _ in range(5)])
        print("  %d GEMM(s) + dW + Adam: one stream %7.1f us, two streams %7.1f us  (hidden: %.1f us)" % (n, s, o, s - o))
    sys.stdout.flush()
_lib.synchronize()
comm.close()
