#!/bin/bash
# One profiling pass of the round on the GPU box (run through gpurun from the repository root):
#   bash tools/profile_round.sh r03
# Bench lines, rocprofv3 kernel traces of the same commands, separate --pmc passes (FETCH_SIZE / WRITE_SIZE for the HBM-side
# traffic, SQ counters for the bf16 GEMM).  Everything lands under gpurun_out/<round>prof/; tools/rocpd_summary.py,
# tools/rocpd_pmc.py and tools/traffic_from_pmc.py turn the databases into the text files committed under profiles/.
R=${1:-r05}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${R}prof
mkdir -p $OUT
run() { echo "== $*" >> $OUT/log.txt; timeout 900 "$@" >> $OUT/log.txt 2>&1 < /dev/null; echo "rc=$?" >> $OUT/log.txt; }

timeout 900 python3 bench.py --steps 20 --warmup 5 < /dev/null > $OUT/benchA.json 2>> $OUT/log.txt
timeout 900 python3 bench.py --workload C < /dev/null > $OUT/benchC.json 2>> $OUT/log.txt
timeout 900 python3 bench.py --workload E < /dev/null > $OUT/benchE.json 2>> $OUT/log.txt
TNN_FORCE_COMM=1 timeout 900 python3 bench.py --no-cpu-baseline < /dev/null > $OUT/benchA_dp_world1.json 2>> $OUT/log.txt
timeout 900 python3 tools/p2p_bench.py < /dev/null > $OUT/p2p_latency.txt 2>> $OUT/log.txt
for w in 2 4 8; do timeout 400 python3 tools/p2p_stress.py --spawn $w --iters 1500 < /dev/null 2>> $OUT/log.txt | grep "^rank" >> $OUT/p2p_stress.txt; done
# in-kernel wall-clock stamps of the data-parallel step's two communicating launches (debug library: make -C tinynn-autograd_amd/csrc trace)
if [ -f tinynn-autograd_amd/lib/libtnn_hip_trace.so ]; then
  TNN_LIB_PATH=$PWD/tinynn-autograd_amd/lib/libtnn_hip_trace.so TNN_FORCE_COMM=1 timeout 200 python3 tools/probes/ar_fused_trace.py < /dev/null 2>> $OUT/log.txt | grep "blocks\|fwd1" > $OUT/dp_step_stamps.txt
fi
timeout 900 python3 tools/probes/dw_adam.py < /dev/null > $OUT/dw_adam_bf16.txt 2>> $OUT/log.txt
SWEEP_SPLITK=1 timeout 900 python3 tools/gemm_sweep.py < /dev/null > $OUT/gemm_f32_sweep.txt 2>> $OUT/log.txt
timeout 900 python3 tools/probes/soak.py < /dev/null > $OUT/soak.txt 2>> $OUT/log.txt
timeout 900 python3 tools/probes/soak_e.py < /dev/null > $OUT/soak_e.txt 2>> $OUT/log.txt
timeout 900 python3 tools/probes/eager_phases.py < /dev/null > $OUT/eager_phases.txt 2>> $OUT/log.txt
# round 4: the skinny bf16 GEMM's variants / ablations / timelines, the box probe, the counter calibration
timeout 900 tools/probes/bin/gemm_bf16_sk_probe < /dev/null > $OUT/gemm_bf16_sk_probe.txt 2>> $OUT/log.txt
timeout 900 python3 -c "
import json, bench
for i in range(3): print(json.dumps(bench.box_probe()))" < /dev/null > $OUT/box_probe.txt 2>> $OUT/log.txt
# round 5: the loop end to end, in-process A/B probes (config E step forms and kernels, fp32 tile / raster), the overlap probe,
# per-launch times of the reference's own net
timeout 900 python3 bench.py --epoch-loop-only < /dev/null > $OUT/epoch_loop.json 2>> $OUT/log.txt
timeout 900 python3 -m tinynn_autograd_amd.examples.mnist_run --trainer --num_ep 3 --seed 0 < /dev/null > $OUT/mnist_run_trainer.txt 2>> $OUT/log.txt
timeout 900 python3 -m tinynn_autograd_amd.examples.mnist_run --num_ep 2 --seed 0 < /dev/null > $OUT/mnist_run_ops.txt 2>> $OUT/log.txt
timeout 900 python3 tools/probes/e_step_ab.py < /dev/null > $OUT/e_step_ab.txt 2>> $OUT/log.txt
# ... and where the forms differ, kernel by kernel and position by position (two traces: the shipped form and the epilogue-transpose form, each against the 25-launch sequence)
for f in default ct; do
  AB_FORMS=$f,long AB_ROUNDS=4 timeout 900 rocprofv3 --kernel-trace -d $OUT/abtrace_$f -o ab -- python3 tools/probes/e_step_ab.py >> $OUT/log.txt 2>&1 < /dev/null
  db=$(find $OUT/abtrace_$f -name "*.db" | head -1)
  [ -n "$db" ] && { echo "== AB_FORMS=$f,long (the 'prep' form below is '$f')"; python3 tools/probes/e_step_ab_kernels.py $db; } >> $OUT/e_step_ab_kernels.txt 2>> $OUT/log.txt
done
timeout 900 python3 tools/probes/e_kernels_ab.py < /dev/null > $OUT/e_kernels_ab.txt 2>> $OUT/log.txt
timeout 900 python3 tools/probes/gemm_f32_cfg_ab.py < /dev/null > $OUT/gemm_f32_cfg_ab.txt 2>> $OUT/log.txt
timeout 300 python3 tools/probes/e_overlap_probe.py < /dev/null 2>> $OUT/log.txt | grep -v "version\|Hostname\|Librccl" > $OUT/e_overlap_probe.txt
timeout 900 python3 tools/probes/exnet_launches.py < /dev/null > $OUT/exnet_launches.txt 2>> $OUT/log.txt
timeout 900 python3 tools/probes/gemm_f32_data_ab.py < /dev/null > $OUT/gemm_f32_data_ab.txt 2>> $OUT/log.txt
# the bf16 dW + Adam launch taken apart: K sweep / bf16 copies / stand-alone optimizer on the same box; row stride A/B
timeout 600 python3 tools/probes/dw_adam_k.py < /dev/null 2>> $OUT/log.txt | grep "^W " > $OUT/dw_adam_k.txt
timeout 600 python3 tools/probes/dw_adam_stride_ab.py < /dev/null 2>> $OUT/log.txt | grep "bf16 copies\|median" > $OUT/dw_adam_stride_ab.txt
mkdir -p $OUT/cal
timeout 900 tools/probes/bin/fetch_calibration < /dev/null > $OUT/cal/known.txt 2>> $OUT/log.txt
run rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/cal/fetch -o cal -- tools/probes/bin/fetch_calibration
run rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/cal/write -o cal -- tools/probes/bin/fetch_calibration
timeout 900 python3 tools/fetch_calibration.py $OUT/cal/known.txt $(find $OUT/cal/fetch -name "*.db" | head -1) $(find $OUT/cal/write -name "*.db" | head -1) < /dev/null > $OUT/fetch_calibration.txt 2>> $OUT/log.txt
TNN_HOST_COMPILED=0 timeout 900 python3 tools/probes/eager_phases.py < /dev/null >> $OUT/eager_phases.txt 2>> $OUT/log.txt

run rocprofv3 --kernel-trace --stats -d $OUT/ktA -o A -- python3 bench.py --steps 20 --warmup 5
run rocprofv3 --kernel-trace --stats -d $OUT/ktAstep -o Astep -- python3 bench.py --no-extras --steps 2000 --warmup 64
run rocprofv3 --kernel-trace --stats -d $OUT/ktC -o C -- python3 bench.py --workload C --no-cpu-baseline
run rocprofv3 --kernel-trace --stats -d $OUT/ktE -o E -- python3 bench.py --workload E
# the data-parallel step at world 1 (RCCL leg first, then the peer-to-peer leg) and the single-GPU step: duration + gap per launch
TNN_FORCE_COMM=1 timeout 900 rocprofv3 --kernel-trace -d $OUT/ktDP -o dp -- python3 bench.py --no-extras --no-cpu-baseline --steps 2000 --warmup 64 >> $OUT/log.txt 2>&1 < /dev/null

run rocprofv3 --kernel-trace -d $OUT/kt256 -o r256 -- python3 bench.py --rows 256 --no-extras --no-cpu-baseline --steps 2000 --warmup 64

run rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcA_fetch -o A -- python3 bench.py --no-extras --steps 200 --warmup 20
run rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcA_write -o A -- python3 bench.py --no-extras --steps 200 --warmup 20
run rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcC_fetch -o C -- python3 bench.py --workload C --no-extras
run rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcC_write -o C -- python3 bench.py --workload C --no-extras
run rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcE_fetch -o E -- python3 bench.py --workload E --no-extras
run rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcE_write -o E -- python3 bench.py --workload E --no-extras
run rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmcbf_sq -o bf -- python3 tools/gemm_bf16_sweep.py
run rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmcg32_sq -o g32 -- python3 tools/gemm_pmc_driver.py
run rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmcsk_sq -o sk -- tools/probes/bin/gemm_bf16_sk_probe

for d in ktA ktAstep ktC ktE; do
    db=$(find $OUT/$d -name "*.db" | head -1)
    [ -n "$db" ] && python3 tools/rocpd_summary.py $db > $OUT/${d}_kernel_stats.txt 2>> $OUT/log.txt
done
db=$(find $OUT/ktDP -name "*.db" | head -1)
if [ -n "$db" ]; then for f in 0.15 0.3 0.5 0.7 0.85; do python3 tools/step_timeline.py $db --frac $f >> $OUT/dp_world1_timeline.txt; done; fi
db=$(find $OUT/ktAstep -name "*.db" | head -1)
[ -n "$db" ] && python3 tools/step_timeline.py $db --frac 0.5 > $OUT/stepA_timeline.txt 2>> $OUT/log.txt
db=$(find $OUT/ktC -name "*.db" | head -1)
[ -n "$db" ] && python3 tools/step_timeline.py $db --frac 0.3 > $OUT/stepC_timeline.txt 2>> $OUT/log.txt
db=$(find $OUT/kt256 -name "*.db" | head -1)
[ -n "$db" ] && python3 tools/step_timeline.py $db --frac 0.5 > $OUT/step256_timeline.txt 2>> $OUT/log.txt
for d in pmcA_fetch pmcA_write pmcC_fetch pmcC_write pmcE_fetch pmcE_write pmcbf_sq pmcg32_sq pmcsk_sq; do
    db=$(find $OUT/$d -name "*.db" | head -1)
    [ -n "$db" ] && python3 tools/rocpd_pmc.py $db > $OUT/${d}.txt 2>> $OUT/log.txt
done
fa=$(find $OUT/pmcA_fetch -name "*.db" | head -1); wa=$(find $OUT/pmcA_write -name "*.db" | head -1)
fc=$(find $OUT/pmcC_fetch -name "*.db" | head -1); wc=$(find $OUT/pmcC_write -name "*.db" | head -1)
fe=$(find $OUT/pmcE_fetch -name "*.db" | head -1); we=$(find $OUT/pmcE_write -name "*.db" | head -1)
timeout 900 python3 tools/traffic_from_pmc.py --round $R A:$fa:$wa C:$fc:$wc E:$fe:$we < /dev/null > $OUT/traffic.json 2>> $OUT/log.txt
# the databases themselves are large: keep only the summaries in what gpurun merges back
find $OUT -name "*.db" -size +20M -delete
tail -5 $OUT/log.txt
ls -la $OUT
