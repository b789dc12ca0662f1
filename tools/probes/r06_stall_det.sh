#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for i in 1 2 3 4 5 6; do echo "# ---- process $i"; timeout 200 python3 tools/probes/epoch_stall_detector.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids"; done
} > gpurun_out/r06/epoch_stall_detector.txt 2>&1
cat gpurun_out/r06/epoch_stall_detector.txt
