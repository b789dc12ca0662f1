#!/bin/bash
# world-1 data-parallel step on both transports (bench.py's collectives object), one line
mkdir -p gpurun_out/dp
TNN_FORCE_COMM=1 timeout 300 python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/dp/bench.txt 2>&1
python3 - <<PY
import json
for l in open("gpurun_out/dp/bench.txt"):
    if l.startswith("{"):
        d = json.loads(l); c = d["config"]["collectives"]
        print("rccl", c["rccl"]["ms_per_step"], "p2p", c["xgmi_p2p"]["ms_per_step"], "verified", c["xgmi_p2p"]["verified_after_run"],
              "single", d.get("single_gpu_bs128", {}).get("ms_per_step"))
PY
