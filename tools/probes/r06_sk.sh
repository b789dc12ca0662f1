#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 tools/probes/bin/gemm_bf16_sk_probe > gpurun_out/r06/gemm_bf16_sk_probe.txt 2>&1
grep -i "128x128\|symmetric hand-off\|baseline\|no exchange\|MFMA-only" gpurun_out/r06/gemm_bf16_sk_probe.txt
rm -rf gpurun_out/r06/pmcsk
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/r06/pmcsk -o sk -- tools/probes/bin/gemm_bf16_sk_probe > gpurun_out/r06/pmcsk.log 2>&1
db=$(find gpurun_out/r06/pmcsk -name "*.db" | head -1)
python3 tools/rocpd_pmc.py $db > gpurun_out/r06/gemm_bf16_sk_pmc.txt 2>&1
rm -rf gpurun_out/r06/pmcsk
grep -c "kernel" gpurun_out/r06/gemm_bf16_sk_pmc.txt
ls /sys/class/drm/ 2>&1 | head; ls /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -3
timeout 600 python3 tools/probes/epoch_stall_clocks.py > gpurun_out/r06/epoch_stall_clocks.txt 2> gpurun_out/r06/epoch_stall_clocks.err
head -40 gpurun_out/r06/epoch_stall_clocks.txt; wc -l gpurun_out/r06/epoch_stall_clocks.txt; tail -5 gpurun_out/r06/epoch_stall_clocks.err
