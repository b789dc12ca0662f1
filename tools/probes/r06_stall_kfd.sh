#!/bin/bash
# the paused epoch against KFD's eviction counter and the kernel's page-migration counters (tools/probes/epoch_stall_kfd.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
ls -la /sys/class/kfd/kfd/proc/ 2>&1 | head -5
for pre in E none; do
  echo "# ===================================================================== PRE=$pre"
  PRE=$pre timeout 300 python3 tools/probes/epoch_stall_kfd.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl"
done > gpurun_out/r06/epoch_stall_kfd.txt 2>&1
head -c 6000 gpurun_out/r06/epoch_stall_kfd.txt
