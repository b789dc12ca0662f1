#!/usr/bin/env python3
"""Post-processing of a rocprofv3 --kernel-trace --hsa-trace --hip-runtime-trace run of epoch_stall_ab.py: for every idle gap > 20 ms
on the GPU's timeline, every HSA / HIP API call (any thread) that overlaps it and lasts > 0.3 ms, and every call that STARTS in the
2 ms before the gap opens.
    python3 tools/probes/epoch_stall_api.py <dir>"""
import csv
import glob
import os
import sys

d = sys.argv[1]


def load(pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        out.extend(csv.DictReader(open(f)))
    return out


kern = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in load("*kernel_trace.csv"))
api = []
for pat, dom in (("*hsa_api_trace.csv", "hsa"), ("*hip_api_trace.csv", "hip")):
    for r in load(pat):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), dom, r["Function"], r.get("Thread_Id", "?")))
api.sort()
t0 = kern[0][0]
print("# %d kernels, %d API calls (%s)" % (len(kern), len(api), ", ".join("%s %d" % (k, sum(1 for a in api if a[2] == k)) for k in ("hsa", "hip"))))
threads = sorted(set(a[4] for a in api))
print("# threads making API calls: %s" % " ".join(threads))
busy_end, prev = kern[0][1], kern[0]
for r in kern[1:]:
    gap = r[0] - busy_end
    if gap > 20_000_000:
        print("gap %.2f ms at t = %.3f .. %.3f ms   after [%s]  before [%s]" % (gap / 1e6, (busy_end - t0) / 1e6, (r[0] - t0) / 1e6, prev[2], r[2]))
        print("   calls overlapping the gap that last > 0.3 ms:")
        for a in api:
            if a[0] < r[0] and a[1] > busy_end and a[1] - a[0] > 300_000:
                print("      %-4s %-44s thread %-8s t = %9.3f .. %9.3f ms (%.2f ms)" % (a[2], a[3], a[4], (a[0] - t0) / 1e6, (a[1] - t0) / 1e6, (a[1] - a[0]) / 1e6))
        print("   calls that start in the 2 ms before the gap opens or the first 1 ms inside it (at most 60):")
        n = 0
        for a in api:
            if busy_end - 2_000_000 <= a[0] <= busy_end + 1_000_000:
                n += 1
                if n <= 60:
                    print("      %-4s %-44s thread %-8s t = %9.3f (%.3f ms)" % (a[2], a[3], a[4], (a[0] - t0) / 1e6, (a[1] - a[0]) / 1e6))
        print("   calls that END in the last 1 ms of the gap or the 1 ms after it (at most 40):")
        n = 0
        for a in api:
            if r[0] - 1_000_000 <= a[1] <= r[0] + 1_000_000:
                n += 1
                if n <= 40:
                    print("      %-4s %-44s thread %-8s t = %9.3f .. %9.3f (%.3f ms)" % (a[2], a[3], a[4], (a[0] - t0) / 1e6, (a[1] - t0) / 1e6, (a[1] - a[0]) / 1e6))
    if r[1] > busy_end:
        busy_end, prev = r[1], r
