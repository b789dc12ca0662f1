#!/bin/bash
# the row-blocked merged head launch (> 128 rows): parity, per-role stamps (trace library), per-launch times, data-parallel world 1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x > gpurun_out/r06/rb_pytest.log 2>&1; tail -3 gpurun_out/r06/rb_pytest.log
for r in 1024 512; do echo "== rows $r"; TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so timeout 300 python3 tools/probes/step_stamps.py $r 2>&1 | grep -A7 "^head"; done
for r in 256 512 1024; do timeout 300 python3 bench.py --rows $r --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print($r, d['ms_per_step'], d['roofline'].get('per_launch_us'))"; done
TNN_FORCE_COMM=1 timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dp world 1:', d['ms_per_step'], {k: (v.get('ms_per_step') if isinstance(v, dict) else v) for k, v in d.get('dp_world1_batch_sizes', {}).items() if k != 'note'})"
