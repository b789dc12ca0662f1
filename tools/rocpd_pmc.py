#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 --pmc result database (rocpd SQLite)."""
import re
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
        # columns typically: ... kernel_name / name, counter_name, value, dispatch_id, start, end
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        q = ("select %s, counter_name, avg(value), count(*), avg(end-start) from counters_collection "
             "group by %s, counter_name order by %s, counter_name" % (name_col, name_col, name_col))
        print("# %s" % path)
        last = None
        for kname, cname, val, n, dur in db.execute(q):
            k = re.sub(r"\(anonymous namespace\)::|^void ", "", kname)[:78]
            if k != last:
                print("%s   (n=%d, avg %.1f us)" % (k, n, (dur or 0) / 1e3))
                last = k
            print("    %-28s %16.1f" % (cname, val))


if __name__ == "__main__":
    main()
