#!/bin/bash
# kernel trace of the world-1 data-parallel step (bench.py, forced communicator) -> per-position timeline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp; rm -rf gpurun_out/dp/kt
TNN_FORCE_COMM=1 timeout 400 rocprofv3 --kernel-trace -d gpurun_out/dp/kt -o dp -- python3 bench.py --no-extras --no-cpu-baseline --steps 2000 --warmup 64 > gpurun_out/dp/kt.log 2>&1
python3 - <<PY
import json
for l in open("gpurun_out/dp/kt.log"):
    if l.startswith("{"):
        d = json.loads(l); c = d["config"]["collectives"]
        print("under the profiler: rccl", c["rccl"]["ms_per_step"], "p2p", c["xgmi_p2p"]["ms_per_step"], "verified", c["xgmi_p2p"]["verified_after_run"])
PY
if [ -f gpurun_out/dp/kt/dp_results.db ]; then python3 tools/step_timeline.py gpurun_out/dp/kt/dp_results.db --frac 0.8 | cut -c1-150; fi
rm -rf gpurun_out/dp/kt
