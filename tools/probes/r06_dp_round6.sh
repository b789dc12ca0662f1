#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for rows in 128 1024; do
  rm -rf gpurun_out/r06/kt$rows
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r06/kt$rows -o t -- python3 tools/probes/dp_vs_single_trace.py $rows 40 > gpurun_out/r06/kt$rows.log 2>&1
  db=$(find gpurun_out/r06/kt$rows -name "*_results.db" | head -1)
  echo "=== rows $rows, single GPU"; python3 tools/step_timeline.py $db --frac 0.2 --take 2000 | cut -c1-130
  echo "=== rows $rows, data parallel at world 1"; python3 tools/step_timeline.py $db --frac 0.7 --take 2000 | cut -c1-130
  rm -rf gpurun_out/r06/kt$rows
done > gpurun_out/r06/dp_vs_single_timeline.txt 2>&1
cat gpurun_out/r06/dp_vs_single_timeline.txt
