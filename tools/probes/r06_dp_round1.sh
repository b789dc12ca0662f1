#!/bin/bash
# round 6, first GPU visit: the data-parallel step's new form (deferred statistics exchange, own-slice Adam in stage B)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1700 python3 -m pytest tests/test_gpu_p2p.py tests/test_gpu_rccl.py -x -q > gpurun_out/r06/pytest_dp.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/pytest_dp.log
tail -5 gpurun_out/r06/pytest_dp.log
for x in 1 0; do
  TNN_DP_XCHG=$x TNN_FORCE_COMM=1 timeout 300 python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/r06/dp_quick_xchg$x.txt 2>&1
  python3 - <<PY
import json
for l in open("gpurun_out/r06/dp_quick_xchg$x.txt"):
    if l.startswith("{"):
        d = json.loads(l); c = d["config"]["collectives"]
        print("TNN_DP_XCHG=$x rccl", c["rccl"]["ms_per_step"], "p2p", c["xgmi_p2p"]["ms_per_step"], "verified", c["xgmi_p2p"]["verified_after_run"],
              "single", d.get("single_gpu_bs128", {}).get("ms_per_step"))
PY
done
for x in 1 0; do
  echo "== stamps, TNN_DP_XCHG=$x"
  TNN_DP_XCHG=$x TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so TNN_FORCE_COMM=1 timeout 300 python3 tools/probes/ar_fused_trace.py 2>&1 | tail -25
done > gpurun_out/r06/ar_trace.txt 2>&1
cat gpurun_out/r06/ar_trace.txt
