#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration per access pattern from the two rocprofv3 --pmc passes over
tools/probes/bin/fetch_calibration (known-byte kernels).  Prints a table and, with --json, the factors
tools/traffic_from_pmc.py applies per kernel:  bytes = counter (KiB) x 1024 x factor.
  python tools/fetch_calibration.py <known.txt> <fetch.db> <write.db> [--json out.json]"""
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
    known_path, fetch_db, write_db = sys.argv[1:4]
    known = {}
    for line in open(known_path):
        if line.startswith("KNOWN "):
            _, name, _, rd, _, wr = line.split()
            known[name] = (int(rd), int(wr))
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    factors = {}
    print("# FETCH_SIZE / WRITE_SIZE (rocprofv3, gfx950) against known bytes; counters are KiB per dispatch (average of %d runs each)"
          % max(n for _, n in fetch.values()))
    print("%-28s %14s %14s %8s   %14s %14s %8s" % ("kernel", "bytes read", "FETCH KiB*1024", "factor", "bytes written", "WRITE KiB*1024", "factor"))
    for name, (rd, wr) in known.items():
        fk = [k for k in fetch if k.startswith(name)]
        wk = [k for k in write if k.startswith(name)]
        f = fetch[fk[0]][0] * 1024 if fk else float("nan")
        w = write[wk[0]][0] * 1024 if wk else float("nan")
        ff = rd / f if rd and f else float("nan")
        wf = wr / w if wr and w else float("nan")
        factors[name] = {"fetch_factor": None if ff != ff else round(ff, 4), "write_factor": None if wf != wf else round(wf, 4)}
        print("%-28s %14d %14.0f %8.3f   %14d %14.0f %8.3f" % (name, rd, f, ff, wr, w, wf))
    if "--json" in sys.argv:
        json.dump(factors, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
