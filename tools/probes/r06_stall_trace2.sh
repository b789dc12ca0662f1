#!/bin/bash
# the paused epoch: GPU gaps against every HSA / HIP API call of the process (tools/probes/epoch_stall_api.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export MODE=none
{
echo "# control, no profiler, this box:"
for i in 1 2 3 4; do timeout 200 python3 tools/probes/epoch_stall_ab.py 2>&1 | grep "^MODE"; done
for i in 1 2 3 4 5 6 7 8; do
  rm -rf /tmp/stall_api_$i
  timeout 500 rocprofv3 --kernel-trace --hsa-trace --hip-runtime-trace --output-format csv -d /tmp/stall_api_$i -- python3 tools/probes/epoch_stall_ab.py 2>&1 | grep "^MODE"
  python3 tools/probes/epoch_stall_api.py /tmp/stall_api_$i
done
} > gpurun_out/r06/epoch_stall_api.txt 2>&1
head -c 14000 gpurun_out/r06/epoch_stall_api.txt
