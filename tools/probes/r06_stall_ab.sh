#!/bin/bash
# bisect the paused epoch's precondition (tools/probes/epoch_stall_ab.py), then look for the gap in a kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for i in 1 2 3; do
  timeout 120 python3 bench.py --epoch-loop-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['epoch_loop']
print('bench.py --epoch-loop-only: trainer', d['trainer']['epoch_ms'], 'ops_captured', d['ops_captured']['epoch_ms'], 'ops_eager', d['ops_eager']['epoch_ms'])"
done
for m in none torch_ctx E_no_torch_ctx E none torch_ctx E_no_torch_ctx E; do
  MODE=$m timeout 200 python3 tools/probes/epoch_stall_ab.py 2>&1 | grep "^MODE"
done
} > gpurun_out/r06/epoch_stall_ab.txt 2>&1
cat gpurun_out/r06/epoch_stall_ab.txt
