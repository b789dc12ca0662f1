#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so timeout 300 python3 tools/probes/step_stamps.py > gpurun_out/r06/stepA_stamps.txt 2>&1
cat gpurun_out/r06/stepA_stamps.txt
timeout 600 python3 tools/probes/adam_margin.py > gpurun_out/r06/adam_margin.txt 2>&1; tail -14 gpurun_out/r06/adam_margin.txt
timeout 900 python3 bench.py > gpurun_out/r06/bench_n1.json 2> gpurun_out/r06/bench_n1.err; echo "bench rc $?"
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r06/bench_n1.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
print("dp_world1", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("dp_world1", {}).items()})
e = d.get("config_E", {})
print("config_E ms", e.get("ms_per_step"), "dw_adam", {k: e.get("dw_adam_roofline", {}).get(k) for k in ("us", "frac", "isolated_us", "isolated_frac", "in_step_us_per_layer_last_first", "frac_of_step_time")})
print("gemm_roofline", e.get("gemm_roofline", {}).get("frac"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["best_leg"])
print("ex net dp", d.get("reference_example_net", {}).get("dp_world1"))
PY
