#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3300 python3 -m pytest tests -m gpu -q -x > gpurun_out/r06/pytest_gpu_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/pytest_gpu_full.log
grep -v "^$" gpurun_out/r06/pytest_gpu_full.log | tail -8
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s); timeout 1200 python3 bench.py > gpurun_out/r06/bench_final.json 2> gpurun_out/r06/bench_final.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r06/bench_final.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "exit", d["exit_code"])
print("dp_world1", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("dp_world1", {}).items() if k != "note"})
print("epoch", d["epoch_loop"]["trainer"]["epoch_ms"], d["epoch_loop"]["trainer"]["all_epochs"])
e = d["config_E"]; print("E", e["ms_per_step"], e["dw_adam_roofline"]["us"], e["dw_adam_roofline"]["frac"], "C", d["config_C"]["ms_per_step"], "g4096", d["roofline_gemm4096"]["frac"])
print("roofline", d["roofline"]["frac"], d["roofline"]["per_launch_us"])
PY
