#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/r06/pytest_wide.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06/pytest_wide.log
timeout 900 python3 -m pytest tests/test_gpu_p2p.py -q -x -k "world1 or two_processes or eight_processes or four_processes or tail_of" > gpurun_out/r06/pytest_wide2.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06/pytest_wide2.log
for rnd in 1 2 3; do
  for w in 0 1; do
    TNN_HEAD_DX_WIDE=$w timeout 200 python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TNN_HEAD_DX_WIDE=$w round $rnd  single', d['ms_per_step'], 'parity', d['parity_vs_reference_fixture']['ok'])"
  done
done > gpurun_out/r06/head_dx_wide_ab.txt 2>&1
cat gpurun_out/r06/head_dx_wide_ab.txt
TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so timeout 300 python3 tools/probes/step_stamps.py 2>&1 | sed -n '12,22p'
