#!/usr/bin/env python3
"""Per-position timeline of a repeating launch sequence from a rocprofv3 --kernel-trace database (rocpd SQLite):
finds the period of the steady-state kernel-name sequence (the training step), then prints for every position of the
period the kernel, its average duration and the average gap between the previous kernel's end and its start.

    python tools/step_timeline.py <results.db> [--skip 2000 | --frac 0.25] [--take 6000] > profiles/r03_stepA_timeline.txt

Sum(duration + gap) over one period = the step time the bench measures; the gaps are what launch boundaries cost."""
import re
import sqlite3
import sys


def short(name, width=100):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= width else name[:width - 3] + "..."


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else None
    take = int(sys.argv[sys.argv.index("--take") + 1]) if "--take" in sys.argv else 6000
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    n = len(rows)
    if "--frac" in sys.argv:
        skip = int(n * float(sys.argv[sys.argv.index("--frac") + 1]))
    if skip is None:
        skip = n // 2                      # the middle of the run: the timed region of a bench
    rows = rows[skip:skip + take]
    names = [r[0] for r in rows]
    period = None
    for p in range(1, 64):
        if len(names) > 4 * p and all(names[i] == names[i + p] for i in range(len(names) - p)):
            period = p
            break
    print("# %s: %d dispatches in the database, window [%d, %d)" % (path, n, skip, skip + len(rows)))
    if period is None:
        print("# no repeating launch sequence with a period below 64 found in the window")
        return
    # align the period on the longest gap (the hipGraph boundary, if any, else position 0)
    reps = len(rows) // period - 1
    dur = [0.0] * period
    gap = [0.0] * period
    for k in range(1, reps + 1):
        for j in range(period):
            i = k * period + j
            dur[j] += rows[i][2] - rows[i][1]
            gap[j] += rows[i][1] - rows[i - 1][2]
    print("# period %d launches, averaged over %d periods" % (period, reps))
    print("%-3s %-100s %10s %10s" % ("pos", "kernel", "dur_us", "gap_us"))
    for j in range(period):
        print("%-3d %-100s %10.3f %10.3f" % (j, short(names[period + j]), dur[j] / reps / 1e3, gap[j] / reps / 1e3))
    td, tg = sum(dur) / reps / 1e3, sum(gap) / reps / 1e3
    print("# per period: kernels %.3f us + gaps %.3f us = %.3f us" % (td, tg, td + tg))


if __name__ == "__main__":
    main()
