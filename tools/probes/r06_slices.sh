#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/r06/pytest_slices.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06/pytest_slices.log
timeout 1200 python3 -m pytest tests/test_gpu_p2p.py -q -x -k "world1 or two_processes or eight_processes or four_processes or tail_of" > gpurun_out/r06/pytest_slices2.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06/pytest_slices2.log
for rnd in 1 2 3; do
  for w in 1 0; do
    for rows in 512 1024; do
      TNN_HEAD_RB_SLICES=$w timeout 200 python3 bench.py --no-extras --no-cpu-baseline --rows $rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TNN_HEAD_RB_SLICES=$w (0 = default slicing) rows $rows round $rnd  single', d['ms_per_step'], 'parity', (d.get('parity_vs_reference_fixture') or {}).get('ok'))"
    done
  done
done > gpurun_out/r06/head_rb_slices_ab.txt 2>&1
cat gpurun_out/r06/head_rb_slices_ab.txt
timeout 300 python3 tools/probes/dp_poll_ab.py 1024 1,32 2>&1 | tail -2
