#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for pre in ${STALL_MODES:-E E_hold alloc none light light_nogap}; do
  PRE=$pre timeout 600 python3 tools/probes/epoch_stall_clocks.py > gpurun_out/r06/epoch_stall_clocks_$pre.txt 2> gpurun_out/r06/epoch_stall_clocks_$pre.err
  echo "=== PRE=$pre"; grep "^# mark\|^#   run 0\|distinct\|^#   pp_\|samples over" gpurun_out/r06/epoch_stall_clocks_$pre.txt
done
