#!/usr/bin/env python3
"""Post-processing of a rocprofv3 --kernel-trace (+ --memory-copy-trace) run of epoch_stall_ab.py: where on the GPU's own timeline
is the paused epoch?  Prints every idle gap > 1 ms between consecutive kernels (end -> next start) with the kernels either side,
every kernel or copy longer than 1 ms, and the copies near the gaps.
    python3 tools/probes/epoch_stall_gaps.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

d = sys.argv[1]
kfiles = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
mfiles = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
rows = []
for f in kfiles:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
copies = []
for f in mfiles:
    for r in csv.DictReader(open(f)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))))
copies.sort()
if not rows:
    raise SystemExit("no kernel trace under %s" % d)
t0 = rows[0][0]
print("# %d kernels, %d copies; t = 0 at the first kernel; queues seen: %s" % (len(rows), len(copies), sorted(set(r[3] for r in rows))))
print("# idle gaps > 1 ms on the GPU's timeline (all queues merged):")
busy_end = rows[0][1]
prev = rows[0]
for r in rows[1:]:
    if r[0] - busy_end > 1_000_000:
        near = [c for c in copies if c[0] < r[0] + 2_000_000 and c[1] > busy_end - 2_000_000]
        print("  gap %8.2f ms  at t = %9.3f ms   after [%s] (q %s)   before [%s] (q %s)" % (
            (r[0] - busy_end) / 1e6, (busy_end - t0) / 1e6, prev[2], prev[3], r[2], r[3]))
        for c in near:
            print("      copy %-14s %12s bytes  t = %9.3f .. %9.3f ms (%.3f ms)" % (c[2], c[3], (c[0] - t0) / 1e6, (c[1] - t0) / 1e6, (c[1] - c[0]) / 1e6))
    if r[1] > busy_end:
        busy_end, prev = r[1], r
print("# kernels longer than 1 ms:")
for r in rows:
    if r[1] - r[0] > 1_000_000:
        print("  %8.2f ms  at t = %9.3f ms  [%s] (q %s)" % ((r[1] - r[0]) / 1e6, (r[0] - t0) / 1e6, r[2], r[3]))
print("# copies longer than 1 ms:")
for c in copies:
    if c[1] - c[0] > 1_000_000:
        print("  %8.2f ms  at t = %9.3f ms  %s %s bytes" % ((c[1] - c[0]) / 1e6, (c[0] - t0) / 1e6, c[2], c[3]))
# the epoch graphs: runs of the headline step's kernels; the span of each run of >= 1000 consecutive kernels with gaps < 1 ms
print("# busy stretches (>= 1000 kernels with no idle gap > 1 ms): start, span, kernels, sum of kernel durations")
s, n, busy, last_end = rows[0][0], 0, 0, rows[0][1]
for r in rows:
    if r[0] - last_end > 1_000_000:
        if n >= 1000:
            print("  t = %9.3f ms  span %8.3f ms  %5d kernels  busy %8.3f ms" % ((s - t0) / 1e6, (last_end - s) / 1e6, n, busy / 1e6))
        s, n, busy = r[0], 0, 0
    n += 1
    busy += r[1] - r[0]
    last_end = max(last_end, r[1])
if n >= 1000:
    print("  t = %9.3f ms  span %8.3f ms  %5d kernels  busy %8.3f ms" % ((s - t0) / 1e6, (last_end - s) / 1e6, n, busy / 1e6))
