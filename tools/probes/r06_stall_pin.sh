#!/bin/bash
# is the one-off pause triggered by the runtime PINNING the 149.5 MiB host training set for its first upload (copies >= 128 MiB)?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for i in 1 2 3 4 5 6; do
  timeout 200 python3 tools/probes/epoch_stall_prewarm.py 2>&1 | grep "WARM="
  N_TRAIN=40000 timeout 200 python3 tools/probes/epoch_stall_prewarm.py 2>&1 | grep "WARM="
  GPU_PINNED_MIN_XFER_SIZE=4096 timeout 200 python3 tools/probes/epoch_stall_prewarm.py 2>&1 | grep "WARM="
  PREPIN=1 timeout 200 python3 tools/probes/epoch_stall_prewarm.py 2>&1 | grep "WARM="
done
} > gpurun_out/r06/epoch_stall_pin.txt 2>&1
cat gpurun_out/r06/epoch_stall_pin.txt
