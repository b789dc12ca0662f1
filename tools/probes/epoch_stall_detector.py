#!/usr/bin/env python3
"""WHEN does the one-off pause come, relative to what the process has done?  A detector — the headline step's 64-step hipGraph
(1.3 ms) replayed and synchronised in a loop, every replay timed from the host — runs between a scripted sequence of events
(EVENTS, default below); every replay longer than 5 ms is listed with its time since the process's first GPU call.
   det:<ms>     run the detector for that long                     up:<MiB>   upload a FRESH host array of that size
   reup         upload the last host array again                   alloc:<MiB> device allocation, zeroed, released
   idle:<ms>    sleep                                              d2h:<MiB>  read that much back"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bench         # noqa: E402
from tinynn_autograd_amd import _lib, device_array as da   # noqa: E402

torch.cuda.set_device(0)
t_first = time.time()
lib = _lib.get()
run = bench.FusedRun(bench.WIDTHS_A, 128, "softmax_nll", 64)
lib.stream_sync()
events = os.environ.get("EVENTS", "det:400 up:150 det:400 reup det:200 up:150 det:400 alloc:1024 det:200 up:30 det:200 d2h:150 det:300").split()
print("# setup done at t = %.3f s; events: %s" % (time.time() - t_first, " ".join(events)))
last_host, dev, n_rep, long_ones = None, None, 0, []
for ev in events:
    kind, _, arg = ev.partition(":")
    t_ev = time.time()
    if kind == "det":
        end = t_ev + float(arg) * 1e-3
        while time.time() < end:
            t = time.time()
            run.chunk.launch()
            lib.stream_sync()
            dt = time.time() - t
            n_rep += 1
            if dt > 5e-3:
                long_ones.append((t - t_first, dt, ev))
                print("    PAUSE: a replay of %.2f ms at t = %.4f s (during %s, %.1f ms after it began)" % (dt * 1e3, t - t_first, ev, (t - t_ev) * 1e3))
    elif kind == "up":
        last_host = np.random.default_rng(int(t_ev * 1e3) % 1000).random((int(float(arg) * 2**20) // 4,), dtype=np.float32)
        t_ev = time.time()
        dev = da.asarray(last_host)
        lib.stream_sync()
    elif kind == "reup":
        dev = da.asarray(last_host)
        lib.stream_sync()
    elif kind == "alloc":
        tmp = da.zeros((int(float(arg) * 2**20) // 4,), np.float32)
        lib.stream_sync()
        del tmp
    elif kind == "d2h":
        back = np.asarray(dev)
    elif kind == "idle":
        time.sleep(float(arg) * 1e-3)
    print("  t = %7.4f s  %-10s took %7.2f ms" % (t_ev - t_first, ev, (time.time() - t_ev) * 1e3))
print("# %d detector replays, %d pauses > 5 ms%s" % (n_rep, len(long_ones), "" if long_ones else "   (none)"))
