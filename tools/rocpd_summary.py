#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` result database (rocpd SQLite, the default output of
ROCm 7.2's rocprofv3) as a per-kernel table: calls, total / average / min / max duration, share.

    python tools/rocpd_summary.py gpurun_out/prof_A/A_results.db [--demangle-width 110] > profiles/r01_A.txt
"""
import re
import sqlite3
import sys


def short(name, width):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name if len(name) <= width else name[:width - 3] + "..."


def main():
    path = sys.argv[1]
    width = 110
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    span = db.execute("select min(start), max(end), count(*) from kernels").fetchone()
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# %d dispatches, kernel time %.3f ms, first-start to last-end %.3f ms" % (
        span[2], total / 1e6, (span[1] - span[0]) / 1e6))
    print("%-*s %8s %12s %10s %10s %10s %6s %9s %5s %7s %5s" % (
        width, "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "grid_x", "wg", "lds", "vgpr"))
    for name, calls, tot, avg, mn, mx, gx, wg, lds, vg, ag in rows:
        print("%-*s %8d %12.1f %10.3f %10.3f %10.3f %6.2f %9d %5d %7d %5d" % (
            width, short(name, width), calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total,
            gx, wg, lds, (vg or 0) + (ag or 0)))


if __name__ == "__main__":
    main()
