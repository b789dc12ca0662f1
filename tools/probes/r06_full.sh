#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3300 python3 -m pytest tests -m gpu -q -x > gpurun_out/r06/pytest_gpu_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/pytest_gpu_full.log
grep -v "^$" gpurun_out/r06/pytest_gpu_full.log | tail -40
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
