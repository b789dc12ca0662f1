#!/usr/bin/env python3
"""Every HSA / HIP API call of a traced run that lasts > 3 ms, and every queue creation, in time order, with the GPU's idle gaps > 20 ms.
    python3 tools/probes/epoch_stall_longcalls.py <rocprofv3 output dir>"""
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


kern = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]) for r in load("*kernel_trace.csv"))
t0 = kern[0][0]
ev = []
for pat, dom in (("*hsa_api_trace.csv", "hsa"), ("*hip_api_trace.csv", "hip")):
    for r in load(pat):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - s > 3_000_000 or "queue_create" in r["Function"] or "StreamCreate" in r["Function"]:
            ev.append((s, "%-4s %-46s thread %-6s %8.2f ms" % (dom, r["Function"], r.get("Thread_Id", "?"), (e - s) / 1e6)))
busy_end, prev = kern[0][1], kern[0]
for r in kern[1:]:
    if r[0] - busy_end > 20_000_000:
        ev.append((busy_end, "GPU  idle %.2f ms   after [%s] before [%s]" % ((r[0] - busy_end) / 1e6, prev[2], r[2])))
    if r[1] > busy_end:
        busy_end, prev = r[1], r
for s, text in sorted(ev):
    print("  t = %10.3f ms  %s" % ((s - t0) / 1e6, text))
