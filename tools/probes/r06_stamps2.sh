#!/bin/bash
# in-kernel stamps of the final code: the headline step's four launches, and the data-parallel step's fused first-layer launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export TNN_LIB_PATH=tinynn-autograd_amd/lib/libtnn_hip_trace.so
timeout 300 python3 tools/probes/step_stamps.py > gpurun_out/r06/stepA_stamps_final.txt 2>&1
cat gpurun_out/r06/stepA_stamps_final.txt
{
echo "== stamps of dense_bwd0_allreduce_adam_kernel<4, true> (one-rank communicator, peer-to-peer transport), deferred exchange (default)"
TNN_FORCE_COMM=1 timeout 300 python3 tools/probes/ar_fused_trace.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|RuntimeWarning\|ret = ret"
echo "== the same with TNN_DP_XCHG=0 (statistics exchange at the tail of the forward launch: dense_fwd_head_kernel stamped too)"
TNN_DP_XCHG=0 TNN_FORCE_COMM=1 timeout 300 python3 tools/probes/ar_fused_trace.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|RuntimeWarning\|ret = ret"
} > gpurun_out/r06/dp_step_stamps_final.txt 2>&1
cat gpurun_out/r06/dp_step_stamps_final.txt
