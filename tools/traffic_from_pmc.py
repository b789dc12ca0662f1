#!/usr/bin/env python3
"""profiles/r01_traffic.json from rocprofv3 --pmc passes (rocpd SQLite): per kernel and workload tag, the average
HBM-side bytes per launch.  FETCH_SIZE is reported in KiB and counts a wide coalesced stream at HALF its bytes on gfx950
(MI355X_MICROARCH.md, HBM / rocprofv3 section) -> doubled; WRITE_SIZE (KiB) is taken as is.
  python tools/traffic_from_pmc.py A:fetch.db:write.db C:fetch.db:write.db > profiles/r01_traffic.json"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    q = "select %s, avg(value), count(*) from counters_collection where counter_name = ? group by %s" % (name_col, name_col)
    for kname, val, n in db.execute(q, (counter,)):
        out[re.sub(r"\(anonymous namespace\)::|^void ", "", kname)] = (val, n)
    return out


def main():
    table = {}
    for spec in sys.argv[1:]:
        tag, fetch_db, write_db = spec.split(":")
        fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
        for k in sorted(set(fetch) | set(write)):
            f, n = fetch.get(k, (0.0, 0))
            w, _ = write.get(k, (0.0, 0))
            table.setdefault(k, {})[tag] = {"fetch_bytes": int(round(f * 1024 * 2)), "write_bytes": int(round(w * 1024)),
                                            "launches": int(n)}
    print(json.dumps({"_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                      "`python3 bench.py [--workload C]`, round 1, tools/traffic_from_pmc.py; FETCH_SIZE is reported in KiB "
                      "and read at HALF the bytes of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section) -> "
                      "doubled here; WRITE_SIZE (KiB) taken as is.  Bytes per launch.", "kernels": table}, indent=1))


if __name__ == "__main__":
    main()
