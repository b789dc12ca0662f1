#!/bin/bash
# the sticky word's load moved in front of the tile product / beside the first gradient loads: p2p tests, stamps, dp_world1 on the bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_p2p.py -q -x > gpurun_out/r06/dp_entry_pytest.log 2>&1; tail -3 gpurun_out/r06/dp_entry_pytest.log
TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so TNN_FORCE_COMM=1 timeout 300 python3 tools/probes/ar_fused_trace.py 2>&1 | grep "blocks \|launch:" > gpurun_out/r06/dp_entry_stamps.txt; cat gpurun_out/r06/dp_entry_stamps.txt
for i in 1 2; do
TNN_FORCE_COMM=1 timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dp world 1: value', d['value'], 'ms_per_step', d['ms_per_step'], {k: (v.get('ms_per_step') if isinstance(v, dict) else v) for k, v in d.get('dp_world1_batch_sizes', {}).items() if k != 'note'})"
done
