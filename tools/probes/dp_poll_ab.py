#!/usr/bin/env python3
"""World-1 data-parallel step (peer-to-peer transport) against the pacing of the tagged polls: TNN_P2P_POLL_GAP (s_sleep units of 64
clocks between two polls of a slot group) and TNN_P2P_POLL_FIRST (pause of the polling workgroups of the fused first-layer
backward + all-reduce + Adam launch before their first stage-B poll — their producers are tiles of the same launch, nothing can
have arrived before the product).  One process, the configurations in alternating rounds (a box drifts by a few per cent).
    TNN_FORCE_COMM=1 python3 tools/probes/dp_poll_ab.py [rows]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
import tinynn_autograd_amd as tn  # noqa: E402
from tinynn_autograd_amd import _lib  # noqa: E402
from tinynn_autograd_amd.fused import MLPTrainer  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
configs = [(1, 0), (4, 0), (8, 0), (1, 16), (1, 32), (4, 16), (4, 32), (8, 32), (2, 24), (16, 32)]
if len(sys.argv) > 2:
    configs = [tuple(int(v) for v in c.split(",")) for c in sys.argv[2:]]
os.environ["TNN_FORCE_COMM"] = "1"
widths = [784, 256, 128, 10]
rng = np.random.default_rng(0)
x = tn.asarray((rng.random((rows, 784)) * (rng.random((rows, 784)) < 0.19)).astype(np.float32))
y = tn.asarray(np.eye(10, dtype=np.float32)[rng.integers(0, 10, rows)])
res = {c: [] for c in configs}
for rnd in range(3):
    for gap, first in configs:
        os.environ["TNN_P2P_POLL_GAP"] = str(gap)
        os.environ["TNN_P2P_POLL_FIRST"] = str(first)
        comm = tn.dist.init_from_env()
        t = MLPTrainer(widths, rows, loss="softmax_nll", optimizer="adam", lr=1e-3, comm=comm, force_dp=True)
        g = t.capture_steps([(x, y)] * 64)
        for _ in range(5):
            g.launch()
        _lib.synchronize()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            g.launch()
        _lib.synchronize()
        res[(gap, first)].append((time.perf_counter() - t0) / (n * 64) * 1e6)
        assert not comm.p2p_status()["dead"]
        del g, t
        comm.close()
print("rows %d; us per data-parallel step at world 1 (three alternating rounds), by (poll gap, first pause) in s_sleep units" % rows)
for c in configs:
    print("gap %2d first %2d : %s  median %.2f" % (c[0], c[1], "  ".join("%.2f" % v for v in res[c]), float(np.median(res[c]))))
