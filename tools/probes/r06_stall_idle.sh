#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for i in 1 2 3; do echo "# process $i"; timeout 300 python3 tools/probes/epoch_stall_idle.py 2>&1 | grep "^run"; done
} > gpurun_out/r06/epoch_stall_idle.txt 2>&1
cat gpurun_out/r06/epoch_stall_idle.txt
