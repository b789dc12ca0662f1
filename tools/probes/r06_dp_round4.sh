#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_p2p.py -q -k "sharded_optimizer_step" > gpurun_out/r06/pytest_zero.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/pytest_zero.log
grep -v "^$" gpurun_out/r06/pytest_zero.log | tail -30
timeout 600 python3 tools/probes/dp_poll_ab.py 128 > gpurun_out/r06/dp_poll_ab_128.txt 2>&1; tail -12 gpurun_out/r06/dp_poll_ab_128.txt
timeout 600 python3 tools/probes/dp_poll_ab.py 1024 1,0 4,0 1,32 4,32 > gpurun_out/r06/dp_poll_ab_1024.txt 2>&1; tail -6 gpurun_out/r06/dp_poll_ab_1024.txt
