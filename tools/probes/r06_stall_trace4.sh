#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
echo "# pre-warm A/B (no profiler)"
for cfg in "0 0" "1280 0" "1280 0.5" "0 0" "1280 0" "1280 0.5" "0 0" "1280 0" "1280 0.5" "50000 0" "50000 0"; do
  set -- $cfg
  WARM=$1 SLEEP=$2 timeout 200 python3 tools/probes/epoch_stall_prewarm.py 2>&1 | grep "^WARM"
done
echo "# HSA API trace only (no kernel interception): long calls"
export MODE=none
for i in 1 2 3 4 5; do
  rm -rf /tmp/stall_h_$i
  timeout 500 rocprofv3 --hsa-trace --output-format csv -d /tmp/stall_h_$i -- python3 tools/probes/epoch_stall_ab.py 2>&1 | grep "^MODE"
  python3 - /tmp/stall_h_$i <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*hsa_api_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"]) for r in rows)
t0 = ev[0][0]
print("   %d hsa calls; those > 3 ms:" % len(ev))
for s, e, fn, th in ev:
    if e - s > 3_000_000:
        print("     t = %10.3f ms  %-46s thread %-6s %8.2f ms" % ((s - t0) / 1e6, fn, th, (e - s) / 1e6))
PY
done
} > gpurun_out/r06/epoch_stall_prewarm.txt 2>&1
head -c 12000 gpurun_out/r06/epoch_stall_prewarm.txt
