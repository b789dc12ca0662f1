#!/usr/bin/env python3
"""bench.py — training samples/sec of the MNIST-shape MLP (784-256-128-10, Adam 1e-3), the metric BASELINE.json
names, on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload A|C|E] [--path fused|ops|opsgraph] [--rows R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: zero_grad -> forward -> whole-batch softmax NLL -> backward ->
[all-reduce] -> Adam update (examples/mnist/run.py:79-83).  Inputs are synthetic (MNIST-like sparsity),
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Protocol: `--repeats` (default 5) timed repeats of [W warm-up steps, then K timed steps], each bracketed
by barrier + stream sync + torch.cuda.synchronize(); when K steps take less than `--min-ms` (50 ms) a repeat times
R consecutive K-step segments instead (R chosen from an untimed pilot, equal on all ranks) so the clock never brackets
less than that.  `value` / `ms_per_step` = MEDIAN over the repeats of (max over ranks); min / max are on the line.

What `value` is:
  N = 1   configs[1] of BASELINE.json: bs 128 on one GPU (the configuration the metric is quoted on).
  N > 1   the SAME definition: WEAK scaling — 128 rows per rank, global batch 128 N ("scaling": "weak"; N = 8 is exactly
          configs[3]'s global batch 1024), so `value` at every N divides by the N = 1 `value`.  `speedup_vs_n1` is that
          ratio computed inside the run: `value` / (the 128-row single-GPU step measured on rank 0 of the same run,
          `single_gpu_bs128`).  `--scaling strong` puts `value` on the other curve instead.  Every line also carries
            weak_scaling    128 rows per rank (global batch 128 N)
            strong_scaling  configs[3] taken literally: global batch 1024 split over the N ranks (N = 1: bs 1024 on one
                            GPU = `--workload A --rows 1024`), with its own `speedup_vs_n1` against `single_gpu_bs1024`
            weak_scaling_1024  1024 rows per rank (global batch 1024 N), speedup against `single_gpu_bs1024` — the only one of
                            the three definitions under which north_star's >= 6x at 8 GPUs is arithmetically reachable
                            (DESIGN.md §7 states the three ceilings)
            batch_sizes     (N = 1) the same net at 256 and 512 rows on the one GPU — the per-rank batches of N = 4 / 2
          so all curves can be drawn from the driver's N = 1/2/4/8 lines.  N > 1 lines also carry
            parity_vs_reference_fixture  the trainer + transport against the reference's own per-step losses (traj_A / traj_D)
            parity_at_timed_rows         the step form actually TIMED (its rows per rank) against the single-GPU trainer on the
                                         concatenated global batch, replicas identical
            multi_gpu       RCCL version, hipDeviceCanAccessPeer / link-type matrix, both transports' self-test verdicts before
                            and after the timed runs, us per collective at this N (940,588-B all-reduce, 2-float all-gather)
                            per transport, which transport `value` came from and the rule that chose it
  Transports at N > 1: RCCL (north_star's named transport) is timed FIRST, the xGMI peer-to-peer path second; both are
  reported unconditionally under config.collectives.  `value` is the peer-to-peer run when that transport passed its
  bit-exact self-test before and after the run, no barrier timed out and all replicas hold identical parameters —
  otherwise the RCCL run.  (A rule, not a best-of-two.)

Extra objects on the N = 1 line:
  roofline           config A is launch-latency bound (129 MFLOP = ~1 us of MFMA time): bound "latency"; peak = the
                     launch-floor model (launches/step x 1.45 us dependent-kernel boundary, MI355X guide price table),
                     achieved = the measured step, both as ksteps/s; per-launch HIP-event times of the step's own
                     launches (tnn_mlp_launch_window), gemm_frac = the step's GEMM FLOPs / their event time / 157.3 TF
  roofline_gemm4096  north_star's ">= 50 % of fp32 MFMA roofline" target: the five 512x4096x4096 GEMMs of config C,
                     HIP events on the library stream, peak 157.3 TFLOP/s, traffic from the PMC passes in profiles/
  config_C           whole-step samples/s of configs[2] (4096-4096-4096, bs 512)
  epoch_loop         the reference's LOOP end to end (examples/mnist_run.train: shuffle, device gather, graph capture, 391
                     steps, loss read-back; eval timed separately) on the trainer / captured-ops / eager-ops paths
  dp_world1          the data-parallel step forms behind a one-rank communicator (128 / 1024 rows per rank)
  reference_example_net  the reference's OWN net 784-200-100-70-30-10: bs 128, batch_sizes 256 / 512 / 1024, dp_world1
  paths              the same config-A step on the drop-in Tensor/ops/Model API: eager and captured (tn.capture)
  cpu_baseline       the numpy port of the reference (oracle/ref_nn.py) on this host, all BLAS threads and 1 thread
  box                what THIS box's MFMA pipes (fp32; bf16 with random / zero operands), clocks and HBM (float4 copy) do,
                     probed in ~100 ms after the timed runs; every mfma / hbm roofline object also carries `frac_of_box`


The code lives in the `bench/` package next to this file (common, clock, runners, roofline, cpu, lines, multi_gpu, main); this file is
the entry the driver calls: it puts the repository root on the path and hands over to bench.main.main().
"""

import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bench.main import main          # noqa: E402  (the package directory `bench/` takes precedence over this module's own name)

if __name__ == "__main__":
    sys.exit(main() or 0)
