#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
for hb in 0 1 0 1 host 1 0 1 host 1 1 1; do HEARTBEAT=$hb timeout 200 python3 tools/probes/epoch_stall_heartbeat.py 2>&1 | grep "HEARTBEAT\|beat of\|paused:"; done
} > gpurun_out/r06/epoch_stall_heartbeat.txt 2>&1
head -c 9000 gpurun_out/r06/epoch_stall_heartbeat.txt
