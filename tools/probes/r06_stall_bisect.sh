#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for i in 1 2 3 4 5 6; do
  for v in ${VARIANTS:-base noeval nogather smallgraph nolosses noeval+nogather+nolosses}; do
    VARIANT=$v timeout 200 python3 tools/probes/epoch_stall_bisect.py 2>&1 | grep "^VARIANT"
  done
done
} > gpurun_out/r06/epoch_stall_bisect.txt 2>&1
sort gpurun_out/r06/epoch_stall_bisect.txt
