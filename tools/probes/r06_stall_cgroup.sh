#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for i in 1 2 3 4; do
  echo "# ---------------- default BLAS pool"; timeout 200 python3 tools/probes/epoch_stall_cgroup.py 2>&1 | grep "^#\|^run"
  echo "# ---------------- BLAS_THREADS=8";   BLAS_THREADS=8 timeout 200 python3 tools/probes/epoch_stall_cgroup.py 2>&1 | grep "^# BLAS\|^# dataset\|^run"
done
} > gpurun_out/r06/epoch_stall_cgroup.txt 2>&1
head -c 14000 gpurun_out/r06/epoch_stall_cgroup.txt
