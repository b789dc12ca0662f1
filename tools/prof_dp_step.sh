# Kernel-trace timelines (duration + gap per launch of the step) of the data-parallel step at world 1 (RCCL leg, then the
# peer-to-peer leg) and of the single-GPU step:  gpurun -- bash tools/prof_dp_step.sh
export TMPDIR=/tmp
O=gpurun_out/r3dp
mkdir -p $O
cd $GRAFT_REPO_ROOT
TNN_FORCE_COMM=1 rocprofv3 --kernel-trace -d $O/p2p -o p2p -- python3 bench.py --no-extras --no-cpu-baseline --steps 2000 --warmup 64 > $O/p2p.json 2> $O/p2p.err
rocprofv3 --kernel-trace -d $O/single -o single -- python3 bench.py --no-extras --no-cpu-baseline --steps 2000 --warmup 64 > $O/single.json 2> $O/single.err
for d in p2p single; do db=$(find $O/$d -name "*.db" | head -1); for f in 0.15 0.3 0.5 0.7 0.85; do python3 tools/step_timeline.py $db --frac $f >> $O/${d}_timeline.txt; done; python3 tools/rocpd_summary.py $db > $O/${d}_stats.txt; done
find $O -name "*.db" -delete
