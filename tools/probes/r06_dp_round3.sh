#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_p2p.py -q -k "sharded_optimizer_step" > gpurun_out/r06/pytest_zero.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/pytest_zero.log
grep -v "^$" gpurun_out/r06/pytest_zero.log | tail -30
bash tools/probes/dp_step_trace.sh > gpurun_out/r06/dp_world1_timeline.txt 2>&1
cat gpurun_out/r06/dp_world1_timeline.txt
