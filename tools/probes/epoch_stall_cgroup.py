#!/usr/bin/env python3
"""Is the one-off paused epoch the container's CPU-bandwidth controller THROTTLING this process (every thread of the cgroup stopped
until the next 100 ms period because the period's CPU quota is spent — e.g. by 64 OpenBLAS threads spinning behind the dataset's
x @ teacher)?  Reads the cgroup's cpu.stat (nr_periods, nr_throttled, throttled time) around every epoch of the trainer-path loop.
    BLAS_THREADS=n   limit numpy's BLAS pool to n threads (threadpoolctl) for the whole run"""
import glob
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)


def cg(name_v1, name_v2, need=None):
    """the file of the nearest cgroup (own one first, then its ancestors up to the namespace's root) that exists [and mentions `need`]"""
    own = {}
    for line in open("/proc/self/cgroup"):
        parts = line.strip().split(":", 2)
        if len(parts) == 3:
            own[parts[1]] = parts[2]
    cands = []
    for ctrl, rel in own.items():
        chain = []
        while True:
            chain.append(rel)
            if rel in ("", "/"):
                break
            rel = os.path.dirname(rel)
        for r in chain:
            r = "" if r == "/" else r
            if ctrl == "":
                cands.append("/sys/fs/cgroup%s/%s" % (r, name_v2))
            elif "cpu" in ctrl.split(","):
                cands += ["/sys/fs/cgroup/%s%s/%s" % (ctrl, r, name_v1), "/sys/fs/cgroup/cpu%s/%s" % (r, name_v1)]
    for p in cands:
        try:
            text = open(p).read()
        except OSError:
            continue
        if need is None or need in text:
            return p, text
    return None, ""


def limited_level():
    """directory of the nearest cgroup (own, then ancestors up to the namespace root) whose CPU quota is set"""
    rel = "/"
    for line in open("/proc/self/cgroup"):
        parts = line.strip().split(":", 2)
        if len(parts) == 3 and parts[1] == "":
            rel = parts[2]
    while True:
        d = "/sys/fs/cgroup" + ("" if rel == "/" else rel)
        try:
            quota = open(d + "/cpu.max").read().split()
            if quota and quota[0] != "max":
                return d, "%s us of CPU per %s us period = %.1f CPUs" % (quota[0], quota[1], int(quota[0]) / float(quota[1]))
        except OSError:
            pass
        if rel in ("", "/"):
            break
        rel = os.path.dirname(rel)
    for d in ("/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):          # cgroup v1
        try:
            q, per = int(open(d + "/cpu.cfs_quota_us").read()), int(open(d + "/cpu.cfs_period_us").read())
            if q > 0:
                return d, "%d us of CPU per %d us period = %.1f CPUs" % (q, per, q / float(per))
        except OSError:
            pass
    return None, "no CPU quota found"


LEVEL, QUOTA = limited_level()


def stat():
    d = {}
    if LEVEL:
        for line in open(LEVEL + "/cpu.stat"):
            k, _, v = line.partition(" ")
            d[k] = int(v)
    thr = d.get("throttled_time", 0) / 1e6 if "throttled_time" in d else d.get("throttled_usec", 0) / 1e3   # ms
    return d.get("nr_periods", 0), d.get("nr_throttled", 0), thr


print("# cgroup of this process: %s" % open("/proc/self/cgroup").read().strip().replace("\n", " | "))
print("# CPU quota: %s (%s)" % (QUOTA, LEVEL))
print("# cpus this process may run on: %d" % len(os.sched_getaffinity(0)))
import numpy as np   # noqa: E402
from threadpoolctl import threadpool_info, threadpool_limits   # noqa: E402
if os.environ.get("BLAS_THREADS"):
    threadpool_limits(int(os.environ["BLAS_THREADS"]))
print("# BLAS pool: %s" % ", ".join("%s %d threads" % (i["internal_api"], i["num_threads"]) for i in threadpool_info()))
import torch         # noqa: E402
from tinynn_autograd_amd import _lib                     # noqa: E402
from tinynn_autograd_amd.examples import mnist_run       # noqa: E402

torch.cuda.set_device(0)
lib = _lib.get()
s0 = stat()
t0 = time.time()
(train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
s1 = stat()
print("# dataset generated on the host in %.0f ms: periods +%d, throttled periods +%d, throttled time +%.1f ms" % (
    (time.time() - t0) * 1e3, s1[0] - s0[0], s1[1] - s0[1], s1[2] - s0[2]))


class Stats(list):
    """records the cgroup counters at every epoch's end"""
    def append(self, s):
        s["cg"] = stat()
        list.append(self, s)


for rep in range(2):
    np.random.seed(0)
    stats = Stats()
    before = stat()
    mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 4, 128, 1e-3, stats=stats, trainer=True)
    lib.stream_sync()
    prev = before
    for e, s in enumerate(stats):
        c = s["cg"]
        print("run %d epoch %d: capture %5.1f steps %6.2f ms | periods +%d, throttled periods +%d, throttled time +%.1f ms%s" % (
            rep, e, s["capture"] * 1e3, s["steps"] * 1e3, c[0] - prev[0], c[1] - prev[1], c[2] - prev[2],
            "   <-- paused" if s["steps"] > 0.02 or s["capture"] > 0.04 else ""))
        prev = c
