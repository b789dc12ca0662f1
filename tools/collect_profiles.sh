#!/bin/bash
# Copy the summaries of one profiling pass (gpurun_out/<round>prof/, produced by tools/profile_round.sh) into profiles/ under
# the round's prefix:  bash tools/collect_profiles.sh r03
R=${1:-r05}
S=gpurun_out/${R}prof
D=profiles
cp $S/benchA.json $D/${R}_benchA.json
cp $S/benchA_dp_world1.json $D/${R}_benchA_dp_world1.json
cp $S/benchC.json $D/${R}_benchC.json
cp $S/benchE.json $D/${R}_benchE.json
cp $S/ktA_kernel_stats.txt $D/${R}_benchA_kernel_stats.txt
cp $S/ktAstep_kernel_stats.txt $D/${R}_benchA_step_kernel_stats.txt
cp $S/ktC_kernel_stats.txt $D/${R}_benchC_kernel_stats.txt
cp $S/ktE_kernel_stats.txt $D/${R}_benchE_kernel_stats.txt
for w in A C E; do cat $S/pmc${w}_fetch.txt $S/pmc${w}_write.txt > $D/${R}_bench${w}_pmc.txt; done
cat $S/pmcbf_sq.txt $S/pmcsk_sq.txt > $D/${R}_gemm_bf16_pmc.txt
cp $S/gemm_bf16_sk_probe.txt $D/${R}_gemm_bf16_sk_probe.txt
cp $S/box_probe.txt $D/${R}_box_probe.txt
cp $S/gemm_f32_data_ab.txt $D/${R}_gemm_f32_data_ab.txt
cp $S/fetch_calibration.txt $D/${R}_fetch_calibration_rerun.txt
cp $S/pmcg32_sq.txt $D/${R}_gemm_f32_pmc.txt
cp $S/traffic.json $D/${R}_traffic.json
cp $S/dp_world1_timeline.txt $D/${R}_dp_world1_timeline.txt
cp $S/stepA_timeline.txt $D/${R}_stepA_timeline.txt
cp $S/p2p_latency.txt $D/${R}_p2p_latency.txt
[ -f $S/p2p_stress.txt ] && cp $S/p2p_stress.txt $D/${R}_p2p_stress.txt
[ -f $S/dp_step_stamps.txt ] && cp $S/dp_step_stamps.txt $D/${R}_dp_step_stamps.txt
cp $S/dw_adam_bf16.txt $D/${R}_dw_adam_bf16.txt
cp $S/gemm_f32_sweep.txt $D/${R}_gemm_f32_sweep.txt
cp $S/soak.txt $D/${R}_soak.txt
cp $S/soak_e.txt $D/${R}_soak_e.txt
cp $S/step256_timeline.txt $D/${R}_step256_timeline.txt
cp $S/eager_phases.txt $D/${R}_eager_phases.txt
for f in e_step_ab_kernels.txt epoch_loop.json mnist_run_trainer.txt mnist_run_ops.txt e_step_ab.txt e_kernels_ab.txt gemm_f32_cfg_ab.txt e_overlap_probe.txt exnet_launches.txt stepC_timeline.txt dw_adam_k.txt dw_adam_stride_ab.txt; do
  [ -f $S/$f ] && cp $S/$f $D/${R}_$f
done
ls -la $D/${R}_*
