"""Why does the plain bf16 dW product (8192 x 8192 x 512 -> fp32) read 95 us in tools/gemm_bf16_sweep.py and 124 us in
tools/probes/dw_adam.py?  Same entry point; candidates: operand magnitudes (the chip's clock follows the data), a fresh output
buffer per call against one reused buffer, the timing helper.

Measured (late round 4, one box): 94-96 us in a back-to-back eager loop whatever the operand scale and the output policy;
106-121 us when 20 launches are replayed from one hipGraph (bench.events_us) — 20 launches rewriting the SAME 268 MB; bench.py's
config_E.gemm_roofline rotates three operand / output sets and reads 95 us from the same helper (its *_same_operands field
shows the 116).  The slow number is the repeated rewrite of one buffer, not the kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib, bf16
import bench
lib = _lib.get()
M = N = 8192; K = 512
rs = np.random.RandomState(0)
for scale_b in (1.0, 1e-2):
    a = bf16.to_bf16(tn.asarray(rs.uniform(-1, 1, (M, K)).astype(np.float32)))
    b = bf16.to_bf16(tn.asarray((rs.uniform(-1, 1, (N, K)) * scale_b).astype(np.float32)))
    G = tn.zeros((M, N))
    def into_G(): lib.gemm_bf16_nt(M, N, K, a._ptr, K, b._ptr, K, G._ptr, N, _lib.F32, None, 0, 0, None, 0)
    def fresh(): return bf16.gemm_nt(a, b, out_dtype=np.float32)
    for name, fn in (("one reused output", into_G), ("fresh output per call", fresh)):
        for _ in range(3): fn()
        e0, e1 = _lib.Event(), _lib.Event()
        e0.record()
        for _ in range(10): fn()
        e1.record()
        loop_us = e0.elapsed_ms(e1) * 100
        print("B scale %-5g %-24s back-to-back loop %6.1f us   bench.events_us %6.1f us" % (scale_b, name, loop_us, bench.events_us(fn, 20)))
