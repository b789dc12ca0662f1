#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_p2p.py -q -x -k "sharded_optimizer_step or world1 or two_processes or four_processes or tail_of_the_forward" > gpurun_out/r06/pytest_dp5.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/pytest_dp5.log
grep -v "^$" gpurun_out/r06/pytest_dp5.log | tail -30
timeout 600 python3 tools/probes/dp_poll_ab.py 1024 1,32 > gpurun_out/r06/dp_1024.txt 2>&1; tail -3 gpurun_out/r06/dp_1024.txt
timeout 600 python3 tools/probes/dp_poll_ab.py 512 1,32 > gpurun_out/r06/dp_512.txt 2>&1; tail -3 gpurun_out/r06/dp_512.txt
