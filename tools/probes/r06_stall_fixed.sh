#!/bin/bash
# after the fix: bench.py --epoch-loop-only x6 and the full default line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for i in 1 2 3 4 5 6; do
  timeout 120 python3 bench.py --epoch-loop-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['epoch_loop']
print('bench.py --epoch-loop-only: trainer', d['trainer']['epoch_ms'], d['trainer']['all_epochs'], '| ops_captured', d['ops_captured']['epoch_ms'], '| ops_eager', d['ops_eager']['epoch_ms'])"
done
} > gpurun_out/r06/epoch_stall_fixed.txt 2>&1
cat gpurun_out/r06/epoch_stall_fixed.txt
t0=$(date +%s); timeout 1200 python3 bench.py > gpurun_out/r06/bench_fixed.json 2> gpurun_out/r06/bench_fixed.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r06/bench_fixed.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "exit", d["exit_code"], "pool", d.get("host_blas_pool"))
print("epoch", d["epoch_loop"]["trainer"]["epoch_ms"], d["epoch_loop"]["trainer"]["all_epochs"])
c = d["cpu_baseline"]; print("cpu", c["value"], c["best_leg"], {k: c[k]["value"] for k in ("all_threads","eight_threads","single_thread") if k in c}, c.get("cpu_quota"))
PY
