#!/usr/bin/env python3
"""The one-off 35-80 ms epoch of the reference's loop against the KERNEL DRIVER's view of this process: is the pause a KFD queue
eviction (the driver takes every user queue of a process off the hardware while it revalidates memory — after an MMU-notifier
invalidation of host pages the runtime pinned for a copy, a page migration by automatic NUMA balancing, a TTM eviction — and puts
them back later)?  A second process that never touches the GPU samples, about every 0.3 ms:
    /sys/class/kfd/kfd/proc/*/stats_*/evicted_ms      time the process's queues spent evicted, per GPU
    /proc/vmstat                                      numa_pte_updates numa_hint_faults numa_pages_migrated pgmigrate_success
                                                      compact_stall thp_collapse_alloc thp_split_page
and the epochs' wall-clock windows are laid over the samples (PRE as in epoch_stall_clocks.py).
    PRE=E python3 tools/probes/epoch_stall_kfd.py
Sampler mode (internal): epoch_stall_kfd.py --sample <out file> <stop file>"""
import glob
import os
import subprocess
import sys
import time

VM_KEYS = ("numa_pte_updates", "numa_hint_faults", "numa_pages_migrated", "pgmigrate_success", "compact_stall",
           "thp_collapse_alloc", "thp_split_page", "thp_fault_alloc")


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError as e:
        return "?%s" % e.errno


def sample_loop(out_path, stop_path):
    with open(out_path, "w") as out:
        n = 0
        ev = []
        while not os.path.exists(stop_path):
            if n % 200 == 0:
                ev = sorted(glob.glob("/sys/class/kfd/kfd/proc/*/stats_*/evicted_ms"))
            n += 1
            t = time.time()
            vm = {}
            for line in read("/proc/vmstat").splitlines():
                k, _, v = line.partition(" ")
                if k in VM_KEYS:
                    vm[k] = v
            out.write("%.6f %s | %s\n" % (t, " ".join(vm.get(k, "-") for k in VM_KEYS),
                                          " ".join("%s=%s" % ("/".join(p.split("/")[-3:-1]), read(p)) for p in ev)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--sample":
        return sample_loop(sys.argv[2], sys.argv[3])
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, root)
    tmp = os.environ.get("TMPDIR", "/tmp")
    samples, stop = os.path.join(tmp, "tnn_kfd_samples.txt"), os.path.join(tmp, "tnn_kfd_stop")
    for p in (samples, stop):
        if os.path.exists(p):
            os.remove(p)
    print("# /proc/sys/kernel/numa_balancing = %s   transparent_hugepage/enabled = %s   defrag = %s" % (
        read("/proc/sys/kernel/numa_balancing"), read("/sys/kernel/mm/transparent_hugepage/enabled"),
        read("/sys/kernel/mm/transparent_hugepage/defrag")))
    print("# numa nodes online: %s   this process may run on cpus: %d" % (read("/sys/devices/system/node/online"), len(os.sched_getaffinity(0))))
    import numpy as np
    sampler = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sample", samples, stop])   # never touches the GPU
    import torch
    import bench
    from tinynn_autograd_amd import _lib
    from tinynn_autograd_amd.examples import mnist_run
    torch.cuda.set_device(0)
    lib = _lib.get()
    marks = [("start", time.time())]
    pre = os.environ.get("PRE", "E")
    if pre == "E":
        bench.config_e_object(bench.Clock(torch, None, 1))
        marks.append(("config E measured, its buffers released", time.time()))
    print("# kfd proc dirs: %s" % " ".join(glob.glob("/sys/class/kfd/kfd/proc/*")))
    for p in glob.glob("/sys/class/kfd/kfd/proc/*/*"):
        if os.path.isfile(p):
            print("#   %s = %s" % (p, read(p)))
    (train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
    marks.append(("dataset on the device", time.time()))
    epochs = []
    for rep in range(2):
        np.random.seed(0)
        stats = []
        mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 6, 128, 1e-3, stats=stats, trainer=True)
        lib.stream_sync()
        for e, s in enumerate(stats):
            epochs.append((rep, e, s))
        marks.append(("run %d of six epochs done" % rep, time.time()))
    with open(stop, "w") as f:
        f.write("stop")
    sampler.wait(timeout=10)
    t0 = marks[0][1]
    rows = []
    for l in open(samples):
        left, _, right = l.partition("|")
        f = left.split()
        rows.append((float(f[0]), f[1:], right.split()))
    print("# %d samples, %.2f ms apart on average" % (len(rows), (rows[-1][0] - rows[0][0]) / max(len(rows) - 1, 1) * 1e3))
    for name, t in marks:
        print("# mark %-45s t = %.3f s" % (name, t - t0))
    for rep, e, s in epochs:
        w = s.get("wall", (t0, t0))
        inside = [r for r in rows if w[0] <= r[0] <= w[1]]
        delta = ""
        if len(inside) >= 2:
            a, b = inside[0], inside[-1]
            dv = ["%s +%d" % (k, int(y) - int(x)) for k, x, y in zip(VM_KEYS, a[1], b[1]) if x != "-" and int(y) != int(x)]
            de = ["%s %s -> %s" % (x.split("=")[0], x.split("=")[1], y.split("=")[1]) for x, y in zip(a[2], b[2]) if x != y]
            delta = "   | inside: " + (", ".join(dv + de) or "no counter moved")
        print("#   run %d epoch %d  t = %.4f .. %.4f s  steps %.2f ms%s" % (rep, e, w[0] - t0, w[1] - t0, s.get("steps", 0.0) * 1e3, delta))
    print("# every change of evicted_ms (and the vmstat counters at that sample):")
    last = None
    for t, vm, ev in rows:
        if ev != last:
            print("%9.4f  %s | %s" % (t - t0, " ".join(vm), " ".join(ev)))
            last = ev
    print("# vmstat counters: first sample, last sample")
    for r in (rows[0], rows[-1]):
        print("%9.4f  %s" % (r[0] - t0, " ".join("%s=%s" % kv for kv in zip(VM_KEYS, r[1]))))


if __name__ == "__main__":
    main()
