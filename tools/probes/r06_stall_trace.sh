#!/bin/bash
# the paused epoch on the GPU's own timeline: rocprofv3 kernel + memory-copy trace of epoch_stall_ab.py (MODE=none), twice
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export MODE=none
for i in 1 2; do
  rm -rf /tmp/stall_trace_$i
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/stall_trace_$i -- python3 tools/probes/epoch_stall_ab.py 2>&1 | grep "^MODE"
  python3 tools/probes/epoch_stall_gaps.py /tmp/stall_trace_$i
done > gpurun_out/r06/epoch_stall_gaps.txt 2>&1
head -c 8000 gpurun_out/r06/epoch_stall_gaps.txt
