#!/usr/bin/env python3
"""profiles/<round>_traffic.json from rocprofv3 --pmc passes (rocpd SQLite): per kernel and workload tag, the average
HBM-side bytes per launch, and per workload the bytes of ONE training step.  FETCH_SIZE is reported in KiB and counts a
wide coalesced stream at HALF its bytes on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section) -> doubled; WRITE_SIZE
(KiB) is taken as is.  Round 4 calibrated both counters per access pattern on known-byte kernels
(profiles/r04_fetch_calibration.txt: tools/probes/fetch_calibration.hip + tools/fetch_calibration.py): the factor is 2.000
for 16 / 8 / 4-byte streams, 4-byte MFMA-fragment-shaped loads, buffer loads and LDS-DMA alike, WRITE_SIZE 1.000 for every
store width — one factor for every kernel is what the measurement says, so FETCH_FACTOR stays a single constant.
  python tools/traffic_from_pmc.py --round r02 A:fetch.db:write.db C:fetch.db:write.db > profiles/r02_traffic.json
The passes must profile a command that runs NOTHING but training steps (`bench.py --no-extras`): bytes per step =
sum over kernels of (average bytes per launch x launches per step), launches per step = launches / launches of the
kernel that runs exactly once per step (ONCE_PER_STEP)."""
import json
import re
import sqlite3
import sys

# kernels that run exactly once per training step: the optimizer-carrying first-layer backward (config A), the loss kernels
# (configs C / E)
FETCH_FACTOR, WRITE_FACTOR = 2.0, 1.0       # profiles/r04_fetch_calibration.txt
ONCE_PER_STEP = ("dense_bwd0_adam_kernel", "mse_fwd_bwd_kernel", "mse_bf16_kernel")


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
    args = sys.argv[1:]
    rnd = "r02"
    if args and args[0] == "--round":
        rnd, args = args[1], args[2:]
    table, steps = {}, {}
    for spec in args:
        tag, fetch_db, write_db = spec.split(":")
        fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
        for k in sorted(set(fetch) | set(write)):
            f, n = fetch.get(k, (0.0, 0))
            w, _ = write.get(k, (0.0, 0))
            table.setdefault(k, {})[tag] = {"fetch_bytes": int(round(f * 1024 * FETCH_FACTOR)), "write_bytes": int(round(w * 1024 * WRITE_FACTOR)),
                                            "launches": int(n)}
        once = [k for k in table if tag in table[k] and k.startswith(ONCE_PER_STEP)]
        if once:
            n_steps = max(table[k][tag]["launches"] for k in once)
            total, parts = 0.0, {}
            for k, per in table.items():
                if tag not in per or k.startswith("__amd_rocclr"):
                    continue
                share = (per[tag]["fetch_bytes"] + per[tag]["write_bytes"]) * per[tag]["launches"] / float(n_steps)
                total += share
                parts[k[:70]] = int(round(share))
            steps[tag] = {"bytes_per_step": int(round(total)), "steps_profiled": int(n_steps), "by_kernel": parts}
    print(json.dumps({"_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                      "`python3 bench.py --no-extras [--workload C]`, round %s, tools/profile_round.sh + tools/traffic_from_pmc.py; "
                      "FETCH_SIZE is reported in KiB and read at HALF the bytes of a wide coalesced stream on gfx950 "
                      "(MI355X_MICROARCH.md, HBM section) -> doubled here; WRITE_SIZE (KiB) taken as is.  kernels: bytes "
                      "per launch; steps: bytes of one training step over all its kernels." % rnd,
                      "kernels": table, "steps": steps}, indent=1))


if __name__ == "__main__":
    main()
