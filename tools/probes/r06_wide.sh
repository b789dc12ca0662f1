#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "trainer or traj or keep_grads or step_forms" > gpurun_out/r06/pytest_wide.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06/pytest_wide.log
timeout 600 python3 -m pytest tests/test_gpu_p2p.py -q -x -k "world1 or two_processes_five or eight_processes" > gpurun_out/r06/pytest_wide2.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06/pytest_wide2.log
for rnd in 1 2 3; do
  for w in 0 1; do
    TNN_DW0_WIDE=$w TNN_FORCE_COMM=1 timeout 200 python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']['collectives']
print('TNN_DW0_WIDE=$w round $rnd  dp world-1 p2p', c['xgmi_p2p']['ms_per_step'], 'rccl', c['rccl']['ms_per_step'])"
    TNN_DW0_WIDE=$w timeout 200 python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TNN_DW0_WIDE=$w round $rnd  single', d['ms_per_step'], 'parity', d['parity_vs_reference_fixture']['ok'])"
  done
done > gpurun_out/r06/dw0_wide_ab.txt 2>&1
cat gpurun_out/r06/dw0_wide_ab.txt
