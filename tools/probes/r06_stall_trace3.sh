#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export MODE=none
{
for i in 1 2 3; do
  rm -rf /tmp/stall_l_$i
  timeout 500 rocprofv3 --kernel-trace --hsa-trace --hip-runtime-trace --output-format csv -d /tmp/stall_l_$i -- python3 tools/probes/epoch_stall_ab.py 2>&1 | grep "^MODE"
  python3 tools/probes/epoch_stall_longcalls.py /tmp/stall_l_$i
done
} > gpurun_out/r06/epoch_stall_longcalls.txt 2>&1
head -c 9000 gpurun_out/r06/epoch_stall_longcalls.txt
