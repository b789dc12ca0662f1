#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python3 -m pytest tests/test_gpu_p2p.py tests/test_gpu_rccl.py -q > gpurun_out/r06/pytest_dp2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/pytest_dp2.log
grep -v "^$" gpurun_out/r06/pytest_dp2.log | tail -30
