#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for rnd in 1 2 3; do
  for w in 0 8; do
    TNN_SMALL_WAVES=$w timeout 200 python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TNN_SMALL_WAVES=$w round $rnd ms_per_step', d['ms_per_step'], 'parity', d['parity_vs_reference_fixture']['ok'])"
  done
done > gpurun_out/r06/fwd0_waves_ab.txt 2>&1
cat gpurun_out/r06/fwd0_waves_ab.txt
TNN_SMALL_WAVES=8 TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so timeout 300 python3 tools/probes/step_stamps.py 2>&1 | head -12
