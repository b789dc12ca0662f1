#!/bin/bash
# does the one-off pause need the process's SECOND hardware queue (created by the first hipGraphInstantiate)?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for i in 1 2 3 4 5 6 7 8 9 10; do
  echo -n "default            "; WARM=0 SLEEP=0 timeout 200 python3 tools/probes/epoch_stall_prewarm.py 2>&1 | grep "^WARM"
  echo -n "GPU_MAX_HW_QUEUES=1 "; GPU_MAX_HW_QUEUES=1 WARM=0 SLEEP=0 timeout 200 python3 tools/probes/epoch_stall_prewarm.py 2>&1 | grep "^WARM"
done
} > gpurun_out/r06/epoch_stall_queues.txt 2>&1
cat gpurun_out/r06/epoch_stall_queues.txt
