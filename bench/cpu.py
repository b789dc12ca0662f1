"""The CPU baseline leg: the numpy port of the reference (oracle/ref_nn.py) timed on this host — the only place besides the
fixture check where bench touches oracle/."""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

from .common import synth_batches


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(widths, rows, kind, budget_s=8.0, host_pool=None):
    """The numpy port of the reference on this host (bounded sample of the same workload): with every BLAS thread the host
    offers, with 8 and with ONE thread (SURVEY §8d asks for all-threads and one); `value` = the fastest leg.
    host_pool: what utils/host_threads.fit_blas_pool_to_cpu_quota() reported at start — the all-threads leg runs with the pool's
    ORIGINAL size (the one numpy chose from the visible CPUs), whatever the process limited it to since."""
    from oracle import ref_nn                              # the reported baseline, never the measured path
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
    except Exception:                                      # noqa: BLE001
        threadpool_info = threadpool_limits = None

    def leg(limit):
        np.random.seed(0)
        layers = ref_nn.build_mlp(widths)
        opt = ref_nn.Adam(lr=1e-3)
        loss_fn = ref_nn.softmax_nll if kind == "softmax_nll" else ref_nn.squared_error
        x, y = synth_batches(4, rows, widths, kind, 0, 1)
        y = y.astype(np.float64)

        def run():
            for i in range(2):
                ref_nn.train_step(layers, opt, loss_fn, x[i * rows:(i + 1) * rows], y[i * rows:(i + 1) * rows])
            t0, steps = time.perf_counter(), 0
            while True:
                i = steps % 4
                ref_nn.train_step(layers, opt, loss_fn, x[i * rows:(i + 1) * rows], y[i * rows:(i + 1) * rows])
                steps += 1
                el = time.perf_counter() - t0
                if el > budget_s or (steps >= 400 and el > 4.0):
                    return steps, el
        if limit is not None and threadpool_limits is not None:
            with threadpool_limits(limits=limit, user_api="blas"):
                return run()
        return run()

    threads = os.cpu_count()
    blas_name = "?"
    all_limit = None
    if threadpool_info is not None:
        blas = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            threads, blas_name = blas[0]["num_threads"], "%s %s" % (blas[0].get("internal_api"), blas[0].get("version"))
            if host_pool and host_pool.get("blas_threads_before") and host_pool["blas_threads_before"] > threads:
                threads = all_limit = host_pool["blas_threads_before"]
    # every leg gets the same budget; `value` is the host's BEST leg (over-subscribed BLAS threads on 128-row GEMMs are slower
    # than one thread: the all-threads leg alone would understate the CPU), all legs stay on the line with their thread counts
    legs = []
    plan = [("all_threads", all_limit, threads)]
    if threadpool_limits is not None:
        if threads > 8:
            plan.append(("eight_threads", 8, 8))           # SURVEY §6's container measurement ran on 8 vCPUs
        plan.append(("single_thread", 1, 1))
    for name, limit, cores in plan:
        s_l, el_l = leg(limit)
        legs.append({"name": name, "value": round(s_l * rows / el_l, 1), "unit": "samples/s", "cores": cores,
                     "sample": "%d steps in %.1f s with %s" % (s_l, el_l, "every BLAS thread the host offers (%d)" % cores
                                                                if name == "all_threads" else "BLAS limited to %d thread%s" % (limit, "s" if limit > 1 else ""))})
    best = max(legs, key=lambda l: l["value"])
    out = {"value": best["value"], "unit": "samples/s", "cores": best["cores"], "kind": "port", "best_leg": best["name"],
           "cpu_model": cpu_model_name(), "logical_cpus": os.cpu_count(), "numpy": np.__version__, "blas": blas_name,
           "sample": "the same %s step (bs=%d) through oracle/ref_nn.py (float64, the reference's per-edge backward), %s; "
                     "value = the fastest of %d legs (%s)" % ("-".join(map(str, widths)), rows, best["sample"], len(legs),
                                                              ", ".join("%s %.0f" % (l["name"], l["value"]) for l in legs))}
    for l in legs:
        out[l["name"]] = {k: l[k] for k in ("value", "unit", "cores", "sample")}
    if host_pool and host_pool.get("quota_cpus"):
        out["cpu_quota"] = {"cpus_per_period": host_pool["quota_cpus"],
                            "note": "the container's CPU bandwidth quota: a leg with more BLAS threads than this is throttled by the "
                                    "kernel (all_threads is such a leg on a %d-CPU host) — why the 8-thread leg wins" % (os.cpu_count() or 0)}
    return out
