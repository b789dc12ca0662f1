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
"""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import tinynn_autograd_amd as tn                      # noqa: E402
from tinynn_autograd_amd import _lib                  # noqa: E402
from tinynn_autograd_amd import device_array as da    # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense (AMD's 5 PF headline includes 2:1 sparsity)
PEAK_HBM_TBS = 8.0
PEAK_HBM_GBS = PEAK_HBM_TBS * 1e3
LAUNCH_BOUNDARY_US = 1.45             # dependent kernel boundary, same stream (MI355X_MICROARCH.md price table)
WIDTHS_A = [784, 256, 128, 10]
WIDTHS_C = [4096, 4096, 4096]
WIDTHS_E = [8192, 8192, 8192, 8192, 8192]
GLOBAL_BATCH_D = 1024
PROFILE_ROUND = "r06"


# ------------------------------------------------------------------------------------------------ data / nets
def synth_batches(n_batches, rows, widths, kind, rank, world, seed=1234):
    """Global batches of rows*world samples from one seeded stream; this rank keeps its row block."""
    rs = np.random.RandomState(seed)
    xs, ys = [], []
    for _ in range(n_batches):
        x = rs.rand(rows * world, widths[0]).astype(np.float32)
        if kind == "softmax_nll":
            x *= (rs.rand(rows * world, widths[0]) < 0.19)
            y = np.eye(widths[-1], dtype=np.float32)[rs.randint(0, widths[-1], rows * world)]
        else:
            y = x
        sl = slice(rank * rows, (rank + 1) * rows)
        xs.append(x[sl])
        ys.append(y[sl])
    return np.concatenate(xs), np.concatenate(ys)


def build_net(widths):
    from tinynn_autograd_amd.core.layers import Dense, ReLU
    from tinynn_autograd_amd.core.nn import Net
    np.random.seed(0)
    layers = []
    for i in range(len(widths) - 1):
        layers.append(Dense(widths[i + 1], num_in=widths[i]))
        if i < len(widths) - 2:
            layers.append(ReLU())
    return Net(layers)


def gemm_list(widths, rows):
    """(name, transA, transB, M, N, K) of every GEMM in one step: fwd NN, dW TN, dX NT (no dX for layer 1)."""
    out = []
    for l in range(len(widths) - 1):
        out.append(("fwd%d" % l, 0, 0, rows, widths[l + 1], widths[l]))
    for l in reversed(range(len(widths) - 1)):
        out.append(("dW%d" % l, 1, 0, widths[l], widths[l + 1], rows))
        if l > 0:
            out.append(("dX%d" % l, 0, 1, rows, widths[l], widths[l + 1]))
    return out


def step_algorithmic(widths, rows):
    """Algorithmic work of one step: GEMM FLOPs; bytes = GEMM operands + 28 B/param Adam."""
    flops = sum(2.0 * M * N * K for _, _, _, M, N, K in gemm_list(widths, rows))
    gemm_bytes = sum(4 * (M * K + K * N + M * N) for _, _, _, M, N, K in gemm_list(widths, rows))
    n_params = sum(widths[l] * widths[l + 1] + widths[l + 1] for l in range(len(widths) - 1))
    return flops, gemm_bytes, 28 * n_params


# ------------------------------------------------------------------------------------------------ kernel-level timing
def events_us(fn, reps):
    """Average duration of `fn`'s launches: `reps` back-to-back calls replayed from ONE hipGraph, HIP events on the
    library stream around the replay (torch.cuda.Event would watch torch's stream, not this one)."""
    for _ in range(3):
        fn()
    graph = _lib.Graph()
    with graph:
        for _ in range(reps):
            fn()
    graph.launch()
    samples = []
    for _ in range(3):
        e0, e1 = _lib.Event(), _lib.Event()
        e0.record()
        graph.launch()
        e1.record()
        samples.append(e0.elapsed_ms(e1) / reps * 1e3)
    return float(np.median(samples))


def time_gemms(widths, rows, reps=20):
    """Each fp32 GEMM of the step on operands shaped like the step's own (activations uniform in [0, 1), weights
    Xavier-uniform: the MFMA data path's power draw, and with it the sustained clock, depends on the values)."""
    lib = _lib.get()
    rs = np.random.RandomState(7)
    results, tot_flops, tot_us = [], 0.0, 0.0
    for name, ta, tb, M, N, K in gemm_list(widths, rows):
        lim = float(np.sqrt(6.0 / (K + N)))
        a = da.asarray(rs.rand(*((K, M) if ta else (M, K))).astype(np.float32))
        b = da.asarray(rs.uniform(-lim, lim, (N, K) if tb else (K, N)).astype(np.float32))
        c = da.empty((M, N), np.float32)
        lda, ldb = (M if ta else K), (K if tb else N)
        us = events_us(lambda: lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, lda, b._ptr, ldb, 0.0, c._ptr, N, _lib.F32), reps)
        flops = 2.0 * M * N * K
        results.append({"gemm": name, "layout": "NT"[ta] + "NT"[tb], "M": M, "N": N, "K": K,
                        "us": round(us, 3), "tflops": round(flops / us / 1e6, 3)})
        tot_flops += flops
        tot_us += us
    achieved = tot_flops / tot_us / 1e6
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
            "kernel": "gemm_f32_mfma_kernel (v_mfma_f32_32x32x2_f32)",
            "algorithmic_gflop_per_step": round(tot_flops / 1e9, 4),
            "gemm_us_per_step": round(tot_us, 2), "per_gemm": results}


def time_gemms_bf16(widths, rows, reps=10):
    """The bf16 step's GEMMs in the K-contiguous form the bf16 trainer uses (tnn_gemm_bf16_nt)."""
    from tinynn_autograd_amd import bf16
    rs = np.random.RandomState(7)
    results, tot_flops, tot_us = [], 0.0, 0.0
    shapes = []
    for l in range(len(widths) - 1):
        shapes.append(("fwd%d" % l, rows, widths[l + 1], widths[l], np.uint16))
    for l in reversed(range(len(widths) - 1)):
        shapes.append(("dW%d" % l, widths[l], widths[l + 1], rows, np.float32))
        if l > 0:
            shapes.append(("dX%d" % l, rows, widths[l], widths[l + 1], np.uint16))
    cache = {}
    rot = 3          # operand sets per shape, used in turn: like the layers of the step, no call finds its weights in the
                     # 256 MB memory-side cache (one 8192 x 8192 bf16 matrix is 134 MB; a single re-used one would stay there)
    for name, M, N, K, out in shapes:
        key = (M, N, K, out)
        if key not in cache:
            # one host draw per operand; the other sets are device-side rescalings of it (different bits, same cost)
            a0 = da.asarray(rs.uniform(-1, 1, (M, K)).astype(np.float32))
            b0 = da.asarray(rs.uniform(-1, 1, (N, K)).astype(np.float32))
            ops = [(bf16.to_bf16(a0 * (0.75 ** i)), bf16.to_bf16(b0 * (0.75 ** i))) for i in range(rot)]
            del a0, b0
            for A, B in ops:
                bf16.gemm_nt(A, B, out_dtype=out)
            e0, e1 = _lib.Event(), _lib.Event()
            e0.record()
            for i in range(reps):
                A, B = ops[i % rot]
                bf16.gemm_nt(A, B, out_dtype=out)
            e1.record()
            rotating = e0.elapsed_ms(e1) / reps * 1e3
            A, B = ops[0]
            for _ in range(2):
                bf16.gemm_nt(A, B, out_dtype=out)
            e0, e1 = _lib.Event(), _lib.Event()
            e0.record()
            for _ in range(reps):
                bf16.gemm_nt(A, B, out_dtype=out)
            e1.record()
            cache[key] = (rotating, e0.elapsed_ms(e1) / reps * 1e3)
            del ops
        us, us_hot = cache[key]
        flops = 2.0 * M * N * K
        results.append({"gemm": name, "layout": "NT(bf16)", "M": M, "N": N, "K": K, "us": round(us, 2),
                        "tflops": round(flops / us / 1e6, 1), "us_same_operands": round(us_hot, 2),
                        "tflops_same_operands": round(flops / us_hot / 1e6, 1)})
        tot_flops += flops
        tot_us += us
    achieved = tot_flops / tot_us / 1e6
    return {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": None,
            "kernel": "sk::gemm_bf16_sk_kernel (256 x 128 tiles, K split over two workgroups, hand-off inside the launch) for the 512-row products, gemm_bf16_dma_kernel (128 x 128) for the dW shape; v_mfma_f32_32x32x16_bf16, fp32 accumulate, LDS-DMA operand rings",
            "operands": "%d sets per shape used in turn (weights come from HBM as in the step); *_same_operands: one set re-used" % rot,
            "algorithmic_gflop_per_step": round(tot_flops / 1e9, 2), "gemm_us_per_step": round(tot_us, 1),
            "per_gemm": results}


def time_dw_adam_bf16(widths, rows, reps=10):
    """configs[4]'s dominant kernel since Adam moved into the dW epilogues: gemm_bf16_dma_kernel<8, 2, false, true>
    (tnn_gemm_bf16_nt_adam) on the weight-gradient shape.  HBM-bound: per parameter it reads p, m, v (12 B) and writes
    p, m, v, the bf16 copy and its transpose (16 B); the operands add 2 x 2 B x rows / n per element."""
    from tinynn_autograd_amd import bf16
    rs = np.random.RandomState(9)
    M, N, K = widths[0], widths[1], rows
    A = bf16.to_bf16(rs.uniform(-1, 1, (M, K)).astype(np.float32))
    B = bf16.to_bf16((rs.uniform(-1, 1, (N, K)) * 1e-2).astype(np.float32))
    P, Mo, Vo = da.zeros((M, N)), da.zeros((M, N)), da.zeros((M, N))
    W16, WT16 = da.empty((M, N), np.uint16), da.empty((N, M), np.uint16)
    pows = da.asarray(np.array([0.5, 0.5, 0, 0]), dtype=np.float64)
    lib = _lib.get()

    def call():
        lib.gemm_bf16_nt_adam(M, N, K, A._ptr, K, B._ptr, K, None, P._ptr, Mo._ptr, Vo._ptr, W16._ptr, WT16._ptr,
                              1e-3, 0.9, 0.999, 1e-8, pows._ptr)
    us = events_us(call, reps)
    alg = 28.0 * M * N + 2.0 * (M + N) * K
    gbs = alg / us / 1e3
    traffic, src = None, None
    table, path = load_traffic_table()
    if table is not None:
        for name, per in table["kernels"].items():
            if "E" in per and name.startswith("gemm_bf16_dma_kernel<8, 2, false, true>"):
                traffic, src = per["E"]["fetch_bytes"] + per["E"]["write_bytes"], path
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "traffic": traffic,
            **({"traffic_unit": "HBM-side bytes per launch (PMC: FETCH_SIZE x 2 + WRITE_SIZE over bench.py --workload E, %s)" % src} if src else {}),
            "kernel": "gemm_bf16_dma_kernel<8, 2, false, true> (dW = a^T dz with Adam in the epilogue, %d x %d x %d)" % (M, N, K),
            "algorithmic_bytes": int(alg), "us": round(us, 1), "launches_per_step": len(widths) - 1,
            "mfma_tflops": round(2.0 * M * N * K / us / 1e6, 1),
            "model": "28 B per parameter (p, m, v read; p, m, v, bf16 copy, bf16 transpose written) + the operands; the "
                     "gradient itself never leaves the accumulators.  Timed here: the 28-B form (3 of the step's 4 launches); "
                     "the first layer's launch writes no [in, out] bf16 copy (nothing reads it: dX stops at the input) = 26 B"}


def in_step_launch_us(run, positions, reps=4):
    """HIP-event time of single launches INSIDE the step: launch k costs T(the step's launches 0 .. k) - T(launches 0 .. k - 1),
    each prefix replayed `reps` times back to back from one hipGraph (tnn_mlp_launch_window restricts tnn_mlp_step to a window
    of its primitive calls).  Unlike a stand-alone replay of one launch on one operand set, the launch finds the caches as the
    step leaves them: its operands written by the launch in front of it, everything older evicted by the step's own traffic."""
    lib, h = run.trainer._lib, run.trainer._h
    x, y = run.batches[0]
    prefix = {}
    try:
        for k in sorted(set(positions) | set(p + 1 for p in positions)):
            if k == 0:
                prefix[0] = 0.0
                continue
            lib.mlp_launch_window(h, 0, k, None)
            prefix[k] = events_us(lambda: lib.mlp_step(h, x._ptr, y._ptr, run.rows, None), reps)
    finally:
        lib.mlp_launch_window(h, 0, -1, None)
    return [prefix[p + 1] - prefix[p] for p in positions]


def dw_adam_roofline_in_step(run, widths, rows, ms_per_step):
    """config E's roofline object: the dW + Adam launch against 8 TB/s with its time taken INSIDE the step (in_step_launch_us);
    the stand-alone replay on one operand set — which stays in the memory-side cache and reads 5-20 % faster — is kept as
    isolated_*.  `run`: the bf16 FusedRun whose step is being reported."""
    # where the dW + Adam launches sit in the step's launch sequence (csrc/tnn_mlp.cpp mlp16_step_fused): L forward, the
    # loss / dz launch, then per layer, last first: transposes, dX (not for the first layer), dW + Adam; one bias launch
    L, k, dw_pos = len(widths) - 1, len(widths), []
    for l in reversed(range(L)):
        k += 1 + (1 if l > 0 else 0)
        dw_pos.append(k)
        k += 1
    in_step = in_step_launch_us(run, dw_pos) if run.launches_per_step() == k + 1 else None
    roof = time_dw_adam_bf16(widths, rows, reps=6)
    if in_step is None:
        roof["frac_of_step_time"] = round(roof["us"] * L / (ms_per_step * 1e3), 3)
        return roof
    us28 = float(np.mean(in_step[:-1])) if L > 1 else float(in_step[0])
    roof["isolated_us"], roof["isolated_achieved"], roof["isolated_frac"] = roof["us"], roof["achieved"], roof["frac"]
    roof["us"] = round(us28, 1)
    roof["achieved"] = round(roof["algorithmic_bytes"] / us28 / 1e3, 1)
    roof["frac"] = round(roof["achieved"] / PEAK_HBM_GBS, 4)
    roof["mfma_tflops"] = round(2.0 * widths[0] * widths[1] * rows / us28 / 1e6, 1)
    roof["in_step_us_per_layer_last_first"] = [round(v, 1) for v in in_step]
    roof["timed"] = ("inside the %d-launch step: T(launches 0 .. k) - T(launches 0 .. k - 1) with HIP events, the step restricted to "
                     "a prefix of its launches (tnn_mlp_launch_window); us = mean of the 28-byte launches (every layer but the first, "
                     "whose 26-byte launch is the last entry of in_step_us_per_layer_last_first)" % (k + 1))
    roof["frac_of_step_time"] = round(float(np.sum(in_step)) / (ms_per_step * 1e3), 3)
    return roof


def config_e_object(clock, rank=0, world=1, comm=None, force_dp=False):
    """configs[4] on the driver's line: the whole bf16 step (8192-wide x 4, 512 rows per GPU, Adam) — single GPU: Adam in
    the dW epilogues; data-parallel: the sharded-optimizer step (reduce-scatter bf16 dW / Adam on the owned rows /
    all-gather bf16 W, csrc/tnn_mlp.cpp mlp16_step_zero) — plus, on one GPU, its dominant kernel against the HBM roofline
    and its GEMMs against the bf16 MFMA peak."""
    e = FusedRun(WIDTHS_E, 512, "mse", 2, rank, world, comm, force_dp, dtype="bfloat16")
    re = measure(clock, e, 2, 6, 3, 0.0, 512 * world)
    gflop = 755.9
    obj = brief(re, workload="configs[4]: 8192-wide 4-layer MLP, bf16 storage, fp32 accumulate / master weights / Adam "
                             "state, 512 rows per GPU, sum-of-squares/m", n_gpus=world, algorithmic_gflop_per_step_per_gpu=gflop,
                mfma_frac_of_whole_step=round(gflop * 1e9 / (re["ms_per_step"] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4))
    if comm is not None:
        n = e.trainer.n_params
        obj["step_form"] = ("sharded optimizer: per layer reduce-scatter(bf16 dW) -> Adam on the owned rows -> all-gather(bf16 W) on "
                            "the communication stream, overlapping the remaining backward; one small fp32 all-reduce for biases + loss")
        obj["wire_bytes_per_step_per_gpu"] = int(2 * (world - 1) / max(world, 1) * 2 * n)
        obj["collectives_on"] = ("rccl" if getattr(comm, "_rccl", False) else
                                 "xgmi peer-to-peer bulk path (no RCCL communicator: direct exchange over the IPC-mapped regions, "
                                 "%d MiB of staging per source)" % (getattr(comm, "p2p_bulk_bytes", 0) >> 20))
        w16 = np.asarray(e.trainer.weights_bf16())
        crc = int(np.frombuffer(w16.tobytes(), dtype=np.uint32).sum(dtype=np.uint64))
        if world > 1:
            import torch.distributed as dist
            box = [None] * world
            dist.all_gather_object(box, crc)
            obj["replicas_identical_bf16_weights"] = bool(all(c == box[0] for c in box))
    else:
        obj["step_form"] = "single GPU: Adam in the epilogue of every dW GEMM (keep_grads off)"
    if comm is None:
        obj["dw_adam_roofline"] = dw_adam_roofline_in_step(e, WIDTHS_E, 512, re["ms_per_step"])
    del e
    if comm is None:
        g = time_gemms_bf16(WIDTHS_E, 512, reps=6)
        g.pop("per_gemm", None)
        obj["gemm_roofline"] = g
    return obj


def load_traffic_table():
    """HBM-side bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 per the gfx950 correction
    + WRITE_SIZE, separate rocprofv3 --pmc passes of this same command; tools/traffic_from_pmc.py)."""
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_traffic.json" % rnd)
        if os.path.exists(path):
            return json.load(open(path)), os.path.relpath(path, ROOT)
    return None, None


def attach_gemm_traffic(roof, tag):
    table, src = load_traffic_table()
    if table is None:
        return
    total, algorithmic = 0, 0
    for gm in roof["per_gemm"]:
        akc, bkc = gm["layout"][0] == "N", gm["layout"][1] == "T"
        flags = "%s, %s" % ("true" if akc else "false", "true" if bkc else "false")
        hit = None
        for name, per in table["kernels"].items():
            if tag in per and name.startswith("gemm_f32_mfma_kernel<") and (", " + flags + ", true>") in name:
                hit = per[tag]
        if hit is None:
            return
        total += hit["fetch_bytes"] + hit["write_bytes"]
        if gm["layout"] == "TN":
            # in the profiled step the dW launches carry Adam in their epilogue (tnn_gemm_tn_adam): operands + p, m, v read
            # and written, and no gradient store
            algorithmic += 4 * (gm["M"] * gm["K"] + gm["K"] * gm["N"]) + 24 * gm["M"] * gm["N"]
        else:
            algorithmic += 4 * (gm["M"] * gm["K"] + gm["K"] * gm["N"] + gm["M"] * gm["N"])
    roof["traffic"] = int(total)
    roof["traffic_unit"] = ("bytes per step over the step's five GEMM launches, the two dW launches with their Adam epilogue "
                            "(PMC over bench.py --workload C --no-extras, %s)" % src)
    roof["algorithmic_bytes"] = int(algorithmic)


def step_traffic(tag):
    """HBM-side bytes of ONE whole step (every kernel of the step's graph) from the same table, or None."""
    table, src = load_traffic_table()
    if table is None or "steps" not in table or tag not in table["steps"]:
        return None, None
    return int(table["steps"][tag]["bytes_per_step"]), src


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(widths, rows, kind, budget_s=8.0):
    """The numpy port of the reference on this host (bounded sample of the same workload): with every BLAS thread the host
    offers, with 8 and with ONE thread (SURVEY §8d asks for all-threads and one); `value` = the fastest leg."""
    from oracle import ref_nn                              # the reported baseline, never the measured path
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
    except Exception:                                      # noqa: BLE001
        threadpool_info = threadpool_limits = None

    def leg(limit):
        np.random.seed(0)
        layers = ref_nn.build_mlp(widths)
        opt = ref_nn.Adam(lr=1e-3)
        loss_fn = ref_nn.softmax_nll if kind == "softmax_nll" else ref_nn.squared_error
        x, y = synth_batches(4, rows, widths, kind, 0, 1)
        y = y.astype(np.float64)

        def run():
            for i in range(2):
                ref_nn.train_step(layers, opt, loss_fn, x[i * rows:(i + 1) * rows], y[i * rows:(i + 1) * rows])
            t0, steps = time.perf_counter(), 0
            while True:
                i = steps % 4
                ref_nn.train_step(layers, opt, loss_fn, x[i * rows:(i + 1) * rows], y[i * rows:(i + 1) * rows])
                steps += 1
                el = time.perf_counter() - t0
                if el > budget_s or (steps >= 400 and el > 4.0):
                    return steps, el
        if limit is not None and threadpool_limits is not None:
            with threadpool_limits(limits=limit, user_api="blas"):
                return run()
        return run()

    threads = os.cpu_count()
    blas_name = "?"
    if threadpool_info is not None:
        blas = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            threads, blas_name = blas[0]["num_threads"], "%s %s" % (blas[0].get("internal_api"), blas[0].get("version"))
    # every leg gets the same budget; `value` is the host's BEST leg (over-subscribed BLAS threads on 128-row GEMMs are slower
    # than one thread: the all-threads leg alone would understate the CPU), all legs stay on the line with their thread counts
    legs = []
    plan = [("all_threads", None, threads)]
    if threadpool_limits is not None:
        if threads > 8:
            plan.append(("eight_threads", 8, 8))           # SURVEY §6's container measurement ran on 8 vCPUs
        plan.append(("single_thread", 1, 1))
    for name, limit, cores in plan:
        s_l, el_l = leg(limit)
        legs.append({"name": name, "value": round(s_l * rows / el_l, 1), "unit": "samples/s", "cores": cores,
                     "sample": "%d steps in %.1f s with %s" % (s_l, el_l, "every BLAS thread the host offers (%d)" % cores
                                                                if limit is None else "BLAS limited to %d thread%s" % (limit, "s" if limit > 1 else ""))})
    best = max(legs, key=lambda l: l["value"])
    out = {"value": best["value"], "unit": "samples/s", "cores": best["cores"], "kind": "port", "best_leg": best["name"],
           "cpu_model": cpu_model_name(), "logical_cpus": os.cpu_count(), "numpy": np.__version__, "blas": blas_name,
           "sample": "the same %s step (bs=%d) through oracle/ref_nn.py (float64, the reference's per-edge backward), %s; "
                     "value = the fastest of %d legs (%s)" % ("-".join(map(str, widths)), rows, best["sample"], len(legs),
                                                              ", ".join("%s %.0f" % (l["name"], l["value"]) for l in legs))}
    for l in legs:
        out[l["name"]] = {k: l[k] for k in ("value", "unit", "cores", "sample")}
    return out


def all_epochs_object(stats, steady, num_ep, n_train, train_all):
    """Everything from the first shuffle to the last loss — and the same WITHOUT the epochs that carry the one-off GPU-side pause
    (an epoch whose GPU time, `steps`, is more than 3x the steady median; profiles/r06_epoch_stall_clocks.txt: 35-80 ms, once or
    twice per process, 0.3-0.6 s after a sustained power-limited load ended; no sclk / mclk / fclk / socclk level changes with it
    and it does not depend on the large configuration's buffers being released)."""
    med = float(np.median([st["steps"] for st in stats[1:]])) if len(stats) > 1 else float(stats[0]["steps"])
    paused = [i for i, st in enumerate(stats) if st["steps"] > 3.0 * med]
    out = {"value": round(num_ep * n_train / train_all, 1), "train_ms": round(train_all * 1e3, 3), "paused_epochs": paused}
    if paused:
        extra = sum(stats[i]["steps"] - med for i in paused)
        out["without_the_pause"] = {"value": round(num_ep * n_train / (train_all - extra), 1), "train_ms": round((train_all - extra) * 1e3, 3),
                                    "pause_ms": round(extra * 1e3, 3)}
    return out


def epoch_loop_object(headline_value, n_train=50000, n_test=10000, batch_size=128, num_ep=4):
    """The reference's LOOP end to end (examples/mnist/run.py:76-93 + utils/data_iterator.py:22-34), wall clock, through
    this build's counterpart `examples/mnist_run.train`: per epoch np.random.shuffle of the row order, its upload, the
    device gather of inputs and one-hot targets, [graph capture + instantiation], 390 steps of 128 rows + the ragged 80-row
    step, the read-back of the 391 losses — and, timed separately, the evaluation (forward on 10,000 test rows, argmax,
    AccEvaluator).  Three paths: `trainer` (whole-step trainer, the epoch as ONE hipGraph captured in epoch 0 and replayed),
    `ops_captured` (the drop-in Tensor / ops / Model loop body recorded with tn.capture in epoch 1 and replayed), `ops_eager`
    (the same loop body issued op by op from Python: what a user of the reference's loop gets with no opt-in).
    `value` of a path = rows / the MEDIAN wall time of the training part of its steady epochs (the replayed ones on the graph
    paths: epochs >= 1 for the trainer, >= 2 for the recorded op-level loop; every epoch is listed in `epoch_ms`);
    `all_epochs` is everything from the first shuffle to the last loss, captures included."""
    import gc
    from tinynn_autograd_amd.examples import mnist_run
    gc.collect()                                           # (what earlier measurements of this process left behind goes now)
    _lib.synchronize()
    (train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=n_train, n_test=n_test)
    out = {"workload": "%d epochs x %d rows, bs %d (%d full batches + a ragged %d-row batch), eval on %d rows; %s"
                       % (num_ep, n_train, batch_size, n_train // batch_size, n_train % batch_size, n_test, source),
           "unit": "samples/s", "phases_unit": "ms"}
    ms = lambda v: round(v * 1e3, 3)                                               # noqa: E731
    for name, kw in (("trainer", {"trainer": True}), ("ops_captured", {"capture": True}), ("ops_eager", {})):
        np.random.seed(0)
        stats = []
        t0 = time.perf_counter()
        losses, preds, results = mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], num_ep, batch_size, 1e-3,
                                                 stats=stats, **kw)
        wall = time.perf_counter() - t0
        last = stats[-1]
        train_all = sum(st["train"] for st in stats)
        steady = [st["train"] for st in stats[(2 if name == "ops_captured" else 1):]]
        t_steady = float(np.median(steady))
        out[name] = {
            "value": round(n_train / t_steady, 1), "steady_epoch_ms": ms(t_steady),
            "epoch_ms": [ms(st["train"]) for st in stats],
            "phases_last_epoch": {k: ms(last[k]) for k in ("data", "capture", "steps")},
            "phases_per_epoch": {k: [ms(st[k]) for st in stats] for k in ("data", "capture", "steps", "eval")},
            "all_epochs": all_epochs_object(stats, steady, num_ep, n_train, train_all),
            "eval": {"ms": ms(last["eval"]), "value": round(n_test / last["eval"], 1), "accuracy": results[-1]["accuracy"]},
            "wall_s_incl_setup": round(wall, 3),
            "steps_per_epoch": last["n_steps"], "first_loss": round(losses[0], 6), "last_loss": round(losses[-1], 6),
            "frac_of_headline": round(n_train / t_steady / headline_value, 4),
        }
    # the same loop with the reference's OWN example net (examples/mnist/run.py:59-69: hidden widths 200-100-70-30)
    ex = {}
    for name, kw in (("trainer", {"trainer": True}), ("ops_eager", {})):
        np.random.seed(0)
        stats = []
        losses, preds, results = mnist_run.train(train_x, train_y, test_x, test_y, [200, 100, 70, 30], num_ep, batch_size, 1e-3,
                                                 stats=stats, **kw)
        ex[name] = {"value": round(n_train / float(np.median([st["train"] for st in stats[1:]])), 1), "epoch_ms": [ms(st["train"]) for st in stats],
                    "eval_ms": ms(stats[-1]["eval"]), "last_loss": round(losses[-1], 6), "accuracy": results[-1]["accuracy"]}
    out["reference_example_net"] = ex
    out["note"] = ("headline = the timed replay of pre-captured step graphs over resident batches (`value` of this line); this object "
                   "is the loop a user of examples/mnist/run.py runs.  trainer: epoch 0 pays lazy init + trainer creation + the "
                   "capture of the epoch graph (phases_per_epoch.capture[0]); epochs >= 1 replay it (capture 0.0).  One early epoch of "
                   "the first path may carry a one-off 50-80 ms GPU-side pause (a light load following this line's heavy GEMM "
                   "measurements: not a Python collection, not a HIP call — tools/probes/epoch_stall*.py, JOURNAL.md); `value` is the "
                   "median of the steady epochs and every epoch is listed.")
    return out


# ------------------------------------------------------------------------------------------------ measured runs
class Runner(object):
    """Something that can run `count` consecutive training steps from global step index `first` and report the last
    loss.  Every (first, count) range is replayed from hipGraphs — whole n_batches-step chunks where the step index is
    aligned, shorter pre-captured segment graphs for the unaligned head and tail; `prepare` captures whatever a range
    needs BEFORE the clock starts.  Subclasses provide capture_range(off, length) -> object whose launch() returns the
    per-step losses, and eager_step(i) for the graph-less form."""
    n_batches = 1
    chunk = None

    def plan(self, first, count):
        out, i = [], first
        while count > 0:
            off = i % self.n_batches
            length = min(count, self.n_batches - off)
            out.append((off, length))
            i, count = i + length, count - length
        return out

    def prepare(self, first, count):
        if self.chunk is None:
            return
        for off, length in self.plan(first, count):
            if length != self.n_batches and (off, length) not in self.segments:
                self.segments[(off, length)] = self.capture_range(off, length)

    def run(self, first, count):
        last = None
        if self.chunk is None:
            for i in range(first, first + count):
                last = self.eager_step(i)
            return last
        for off, length in self.plan(first, count):
            g = self.chunk if length == self.n_batches else self.segments[(off, length)]
            last = g.launch()[length - 1]
        return last


class FusedRun(Runner):
    """Whole-step trainer (tnn_mlp_*), replayed from hipGraphs of whole steps bound to their resident batches."""

    def __init__(self, widths, rows, kind, n_batches, rank=0, world=1, comm=None, force_dp=False, use_graph=True,
                 dtype=np.float32, seed=1234):
        self.widths, self.rows, self.kind, self.n_batches = widths, rows, kind, n_batches
        self.comm, self.use_graph = comm, use_graph
        x_host, y_host = synth_batches(n_batches, rows, widths, kind, rank, world, seed=seed)
        self.X, self.Y = da.asarray(x_host), da.asarray(y_host)            # resident in HBM before the timed region
        self.batches = [(self.X[i * rows:(i + 1) * rows], self.Y[i * rows:(i + 1) * rows]) for i in range(n_batches)]
        if isinstance(dtype, str):                                         # bf16 trainer (configs[4])
            from tinynn_autograd_amd import bf16
            from tinynn_autograd_amd.fused import MLPTrainer
            self.trainer = MLPTrainer(widths, rows, loss="mse", optimizer="adam", lr=1e-3, dtype=dtype, comm=comm,
                                      force_dp=force_dp)
            # Adam consumes each weight gradient in the epilogue of the GEMM that produces it; the gradient is not also
            # written to the arena (tests/test_gpu_config_e.py: bit-identical parameters and state either way)
            self.trainer.keep_grads(os.environ.get("TNN_BENCH_KEEP_GRADS", "0") == "1")
            np.random.seed(0)
            for l in range(len(widths) - 1):
                a = np.sqrt(6.0 / (widths[l] + widths[l + 1]))
                self.trainer.param_view(l, "w")[...] = da.asarray(
                    np.random.uniform(-a, a, (widths[l], widths[l + 1])).astype(np.float32))
            _lib.get().mlp_sync_params(self.trainer._h)
            X16 = bf16.to_bf16(self.X)
            self.batches = [(X16[i * rows:(i + 1) * rows], X16[i * rows:(i + 1) * rows]) for i in range(n_batches)]
            self.use_graph = False
        else:
            self.trainer = tn.trainer_from_net(build_net(widths), max_rows=rows, loss=kind, optimizer="adam", lr=1e-3,
                                               comm=comm, use_graph=False, force_dp=force_dp)
            if comm is None:
                # single GPU: Adam consumes the weight gradients where they are produced (configs[2]: every dW epilogue;
                # the MNIST net: the first layer's, the only one its fused step would otherwise write without a reader);
                # tests/test_gpu_fullsize.py / parity_suite: bit-identical parameters and state either way
                self.trainer.keep_grads(os.environ.get("TNN_BENCH_KEEP_GRADS", "0") == "1")
        self.chunk, self.segments = None, {}
        self.capture()

    def capture(self):
        """(Re)capture the chunk graph — also after switching the transport under a data-parallel trainer.  With a
        communicator both collectives of every step are captured too (peer-to-peer kernels, or RCCL which supports
        stream capture); if that capture is refused the run falls back to eager data-parallel steps."""
        self.chunk, self.segments = None, {}
        if not self.use_graph or (self.comm is not None and os.environ.get("TNN_DP_GRAPH", "1") == "0"):
            return
        try:
            self.chunk = self.trainer.capture_steps(self.batches)
        except Exception as exc:                          # noqa: BLE001
            if self.comm is None:
                raise
            sys.stderr.write("bench: data-parallel graph capture unavailable (%s); eager steps\n" % exc)

    def capture_range(self, off, length):
        return self.trainer.capture_steps(self.batches[off:off + length])

    def eager_step(self, i):
        return self.trainer.step(*self.batches[i % self.n_batches])

    def params_crc(self):
        return int(np.frombuffer(np.asarray(self.trainer.params).tobytes(), dtype=np.uint32).sum(dtype=np.uint64))

    def launches_per_step(self):
        """Primitive calls (= kernel launches at this size) of the single-GPU step, counted by the library itself."""
        n = __import__("ctypes").c_int(0)
        self.trainer._lib.mlp_step(self.trainer._h, self.batches[0][0]._ptr, self.batches[0][1]._ptr, self.rows, None)
        self.trainer._lib.mlp_launch_window(self.trainer._h, 0, -1, __import__("ctypes").byref(n))
        return n.value

    def per_launch_us(self, reps=200):
        """HIP-event time of each launch of the step on its own (tnn_mlp_launch_window: the step restricted to its
        k-th primitive call, `reps` back-to-back replays from one hipGraph — so every figure still contains one
        dependent-kernel boundary, like inside the real step)."""
        lib, h = self.trainer._lib, self.trainer._h
        x, y = self.batches[0]
        n = self.launches_per_step()
        out = []
        try:
            for k in range(n):
                lib.mlp_launch_window(h, k, 1, None)
                out.append(round(events_us(lambda: lib.mlp_step(h, x._ptr, y._ptr, self.rows, None), reps), 3))
        finally:
            lib.mlp_launch_window(h, 0, -1, None)
        return out


def fixture_check(widths, rows, kind, rank, world, comm, force_dp, use_graph):
    """Parity of the very trainer + transport being timed, against the REFERENCE's own trajectory: when the global batch is
    one the fixtures were captured at (tests/golden/traj_A_adam.npz: bs 128, 20 steps; traj_D_adam.npz: bs 1024, 5 steps
    — written by oracle/gen_golden.py from the imported reference), a fresh trainer is fed the fixture's batches (this
    rank's row block) and its per-step losses are compared with the reference's (rtol 1e-5)."""
    name = {128: "A_adam", 1024: "D_adam"}.get(rows * world) if (kind == "softmax_nll" and widths == WIDTHS_A) else None
    path = os.path.join(ROOT, "tests", "golden", "traj_%s.npz" % name)
    if name is None or not os.path.exists(path):
        return None
    gold = np.load(path)
    cfg = json.loads(str(gold["config"]))
    steps = int(cfg["steps"])
    fr = FusedRun(widths, rows, kind, steps, rank, world, comm, force_dp, use_graph=use_graph, seed=cfg["data_seed"])
    if fr.chunk is not None:
        losses = np.asarray(fr.chunk.launch(), dtype=np.float64)
    else:
        losses = np.array([float(fr.trainer.step(*b)) for b in fr.batches])
    ref = np.asarray(gold["loss"], dtype=np.float64)[:steps]
    err = float(np.max(np.abs(losses - ref) / np.abs(ref)))
    return {"fixture": "tests/golden/traj_%s.npz (the reference's per-step losses)" % name, "steps": steps,
            "max_rel_err": float("%.3g" % err), "rtol": 1e-5, "ok": bool(err <= 1e-5)}


def timed_rows_check(widths, rows, kind, rank, world, comm, force_dp, use_graph, torch, steps=5):
    """Parity of the step form being TIMED (its rows per rank, its transport): a fresh data-parallel trainer runs `steps`
    steps on seeded global batches of rows x world rows (this rank's row block), rank 0 also runs the SINGLE-GPU trainer on
    the whole concatenated batches — the arithmetic the ranks must reproduce (examples/mnist/run.py:79-83 at that batch
    size) — and the per-step losses are compared (rtol 1e-5); replicas must hold identical parameters afterwards."""
    dp = FusedRun(widths, rows, kind, steps, rank, world, comm, force_dp, use_graph=use_graph, seed=4321)
    if dp.chunk is not None:
        losses = np.asarray(dp.chunk.launch(), dtype=np.float64)
    else:
        losses = np.array([float(dp.trainer.step(*b)) for b in dp.batches])
    crc = dp.params_crc()
    same = True
    if world > 1:
        import torch.distributed as dist
        box = [None] * world
        dist.all_gather_object(box, crc)
        same = bool(all(c == box[0] for c in box))
    out = None
    if rank == 0:
        solo = FusedRun(widths, rows * world, kind, steps, 0, 1, None, False, use_graph=False, seed=4321)
        ref = np.array([float(solo.trainer.step(*b)) for b in solo.batches])
        err = float(np.max(np.abs(losses - ref) / np.abs(ref)))
        blocks = (rows + 127) // 128
        out = {"against": "the single-GPU trainer on the concatenated global batch of %d rows, %d steps" % (rows * world, steps),
               "rows_per_rank": rows, "global_batch": rows * world, "max_rel_err": float("%.3g" % err), "rtol": 1e-5,
               "replicas_identical": same, "ok": bool(err <= 1e-5 and same),
               "step_form": "merged 2L - 2 launch data-parallel step, %d block(s) of <= 128 rows per rank" % blocks}
        del solo
    del dp
    return out


def rccl_version_string():
    try:
        v = ctypes.c_int(0)
        lib = ctypes.CDLL("librccl.so.1")
        lib.ncclGetVersion(ctypes.byref(v))
        n = v.value
        return "%d.%d.%d" % (n // 10000, (n // 100) % 100, n % 100)
    except Exception as exc:                              # noqa: BLE001
        return "unavailable (%s)" % type(exc).__name__


def topology_object(torch):
    """hipDeviceCanAccessPeer and hipExtGetLinkTypeAndHopCount for every pair of visible devices (rank 0; no device is
    initialised by either call).  link types: HSA_AMD_LINK_INFO_TYPE_* (0 HyperTransport, 1 QPI, 2 PCIe, 3 InfiniBand, 4 xGMI)."""
    out = {}
    try:
        n = torch.cuda.device_count()
        out["visible_devices"] = n
        out["can_access_peer"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]
        hip = None
        with open("/proc/self/maps") as f:
            for ln in f:
                if "libamdhip64" in ln:
                    hip = ctypes.CDLL(ln.split()[-1])
                    break
        if hip is not None and n > 1:
            lt, hops = [], []
            for i in range(n):
                lt.append([]); hops.append([])
                for j in range(n):
                    a, b = ctypes.c_uint32(0), ctypes.c_uint32(0)
                    rc = hip.hipExtGetLinkTypeAndHopCount(i, j, ctypes.byref(a), ctypes.byref(b)) if i != j else 0
                    lt[-1].append(int(a.value) if (i != j and rc == 0) else None)
                    hops[-1].append(int(b.value) if (i != j and rc == 0) else 0)
            out["link_type"], out["hops"] = lt, hops
            out["link_type_names"] = {"2": "PCIe", "4": "xGMI"}
    except Exception as exc:                              # noqa: BLE001 - diagnostics never cost the line
        out["error"] = "%s: %s" % (type(exc).__name__, exc)
    return out


def collective_selftest(comm, world, rank, all_ranks):
    """Both transports against known answers, voted over the ranks: RCCL — an all-reduce of rank-dependent integers (exact
    in f32) and the 2-float all-gather; the peer-to-peer transport — its own bit-exact self-test."""
    out = {}
    was = bool(getattr(comm, "_p2p", False))
    if getattr(comm, "_rccl", False):
        if was:
            comm.set_p2p(False)
        try:
            n = 235147
            v = da.asarray(((np.arange(n) % 97) + rank + 1).astype(np.float32))
            comm.allreduce(v)
            want = world * (np.arange(n) % 97).astype(np.float64) + world * (world + 1) / 2.0
            ok = bool(np.array_equal(np.asarray(v, dtype=np.float64), want))
            g = comm.allgather(da.asarray(np.array([rank + 0.5, 2.0 * rank], dtype=np.float32)))
            ok = ok and bool(np.array_equal(np.asarray(g), np.array([[r + 0.5, 2.0 * r] for r in range(world)], dtype=np.float32)))
        except Exception as exc:                          # noqa: BLE001
            sys.stderr.write("bench: RCCL self-test raised: %s\n" % exc)
            ok = False
        out["rccl"] = all_ranks(ok)
        if was:
            comm.set_p2p(True)
    if hasattr(comm, "p2p_status"):
        st = comm.p2p_status()
        if st and st["connected"] and not st["dead"]:
            comm.set_p2p(True)
            try:
                ok = bool(comm.p2p_selftest(sizes=(235147, 4099, 2), rounds=2))
            except Exception as exc:                      # noqa: BLE001
                sys.stderr.write("bench: peer-to-peer self-test raised: %s\n" % exc)
                ok = False
            out["xgmi_p2p"] = all_ranks(ok)
            comm.set_p2p(was)
    return out


def collective_latency_table(comm, clock, reps=100):
    """us per collective at this world size, replayed from one hipGraph of `reps` back-to-back calls (max over ranks):
    the 940,588-byte all-reduce of the gradient arena + loss slot (C1) and the 2-float statistics all-gather (C2), per
    transport."""
    lib = _lib.get()
    world = comm.world
    table = {}
    was = bool(getattr(comm, "_p2p", False))
    legs = []
    if getattr(comm, "_rccl", False):
        legs.append(("rccl", False))
    st = comm.p2p_status() if hasattr(comm, "p2p_status") else None
    if st and st["connected"] and not st["dead"]:
        legs.append(("xgmi_p2p", True))
    for name, p2p in legs:
        comm.set_p2p(p2p)
        row = {}
        try:
            buf = da.asarray(np.zeros(235147, np.float32))
            st2, out2 = da.asarray(np.array([1.0, 2.0], np.float32)), da.empty((world, 2), np.float32)
            for key, fn in (("allreduce_940588_B", lambda: comm.allreduce(buf)),
                            ("allgather_2_floats_per_rank", lambda: lib.allgather(st2._ptr, out2._ptr, 2, _lib.F32))):
                g = _lib.Graph()
                with g:
                    for _ in range(reps):
                        fn()
                g.launch()
                clock.fence()
                t0 = time.perf_counter()
                for _ in range(3):
                    g.launch()
                clock.fence()
                row[key] = round(clock.max_over_ranks(time.perf_counter() - t0) / (3 * reps) * 1e6, 2)
                del g
        except Exception as exc:                          # noqa: BLE001 - diagnostics never cost the line
            row["error"] = "%s: %s" % (type(exc).__name__, exc)
        table[name] = row
    comm.set_p2p(was)
    table["unit"] = "us per collective, %d back-to-back calls per hipGraph launch, max over ranks" % reps
    return table


class _OpsGraph(object):
    def __init__(self, captured):
        self.captured = captured

    def launch(self):
        return [t.values for t in self.captured()]


class OpsRun(Runner):
    """The drop-in API path (core/tensor.py:13-171 / core/ops.py:12-384 are the seam): the reference's loop
    body on Tensor / ops / Dense / ReLU / SoftmaxCrossEntropyLoss / Adam / Model — eager (one launch per op issued from
    Python), or recorded with tn.capture and replayed: like the trainer's graphs, one capture covers a run of steps,
    each bound to its own HBM-resident batch (row slices of the resident dataset, utils/data_iterator.py:30-33), so no
    staging copies are needed."""

    def __init__(self, widths, rows, kind, n_batches, rank=0, world=1, comm=None, graph=False):
        from tinynn_autograd_amd.core.losses import SoftmaxCrossEntropyLoss, SquaredErrorLoss
        from tinynn_autograd_amd.core.model import Model
        from tinynn_autograd_amd.core.optimizer import Adam
        from tinynn_autograd_amd.core.tensor import Tensor
        self.n_batches, self.rows, self.segments = n_batches, rows, {}
        x_host, y_host = synth_batches(n_batches, rows, widths, kind, rank, world)
        X, Y = da.asarray(x_host), da.asarray(y_host)
        self.batches = [(X[i * rows:(i + 1) * rows], Y[i * rows:(i + 1) * rows]) for i in range(n_batches)]
        loss_layer = SoftmaxCrossEntropyLoss(comm=comm) if kind == "softmax_nll" else SquaredErrorLoss()
        model = Model(net=build_net(widths), loss=loss_layer, optimizer=Adam(lr=1e-3), comm=comm)
        tbatches = [(Tensor(a), Tensor(b)) for a, b in self.batches]

        def step(i):
            xb, yb = tbatches[i % n_batches]
            model.zero_grad()
            out = loss_layer.loss(model.forward(xb), yb)
            out.backward()
            model.step()
            return out
        self._step = step
        if graph:
            for i in range(2):                                 # real steps first: arena binding, optimizer state
                step(i)
            self.chunk = self.capture_range(0, n_batches)

    def capture_range(self, off, length):
        return _OpsGraph(tn.capture(lambda: [self._step(i) for i in range(off, off + length)], warmup=0))

    def eager_step(self, i):
        return self._step(i).values


class Clock(object):
    """The bench contract's timed region: barrier + stream sync + torch.cuda.synchronize() on both sides, wall clock,
    max over ranks."""

    def __init__(self, torch, comm, world):
        self.torch, self.comm, self.world = torch, comm, world

    def fence(self):
        if self.comm is not None:
            self.comm.barrier()
        _lib.synchronize()                                   # the library's own stream (kernels + RCCL)
        self.torch.cuda.synchronize()                        # device-wide, as the bench contract asks

    def max_over_ranks(self, dt):
        if self.world > 1:
            import torch.distributed as dist
            t = self.torch.tensor([dt], dtype=self.torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    def timed(self, runner, first, warmup, count):
        runner.prepare(first, warmup)
        runner.prepare(first + warmup, count)                # every graph of the timed region exists before the clock
        runner.run(first, warmup)
        self.fence()
        t0 = time.perf_counter()
        last = runner.run(first + warmup, count)
        self.fence()
        return self.max_over_ranks(time.perf_counter() - t0), last


def measure(clock, runner, warmup, steps, repeats, min_ms, rows_global):
    """`repeats` timed repeats of [warmup, R x steps]; R from an untimed pilot so that a repeat lasts >= min_ms."""
    nb = runner.n_batches
    span = lambda r: (warmup + r * steps + nb - 1) // nb * nb           # noqa: E731  chunk-aligned stride per repeat
    pilot, _ = clock.timed(runner, 0, warmup, steps)
    R = max(1, int(math.ceil(1.25 * min_ms * 1e-3 / max(pilot, 1e-9))))   # the pilot pays one-off costs: margin
    R = min(R, 4096)
    first = span(1)
    per_step, last = [], None
    for _ in range(repeats):
        dt, last = clock.timed(runner, first, warmup, R * steps)
        per_step.append(dt / (R * steps))
        first += span(R)
    med = float(np.median(per_step))
    return {"ms_per_step": med * 1e3, "value": rows_global / med, "min_ms_per_step": min(per_step) * 1e3,
            "max_ms_per_step": max(per_step) * 1e3, "repeats": repeats, "segments_per_repeat": R,
            "timed_steps_per_repeat": R * steps, "final_loss": float(last)}


def brief(res, **extra):
    out = {"value": round(res["value"], 1), "unit": "samples/s", "ms_per_step": round(res["ms_per_step"], 5),
           "min_ms_per_step": round(res["min_ms_per_step"], 5), "max_ms_per_step": round(res["max_ms_per_step"], 5),
           "final_loss": round(res["final_loss"], 6)}
    out.update(extra)
    return out


# ------------------------------------------------------------------------------------------------ self-launch
def self_launch(n, argv):
    """`python3 bench.py --gpus N` without a launcher: THIS process never touches the GPU; it starts N fresh children of
    the same command, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them,
    rendezvous on 127.0.0.1), relays rank 0's single JSON line and exits non-zero as soon as any child does.  Children are
    ended by their exact PIDs only."""
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    limit = float(os.environ.get("TNN_BENCH_LAUNCH_TIMEOUT_S", "1500"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TNN_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    box = {"out": b""}

    def drain():
        box["out"] = procs[0].stdout.read()
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()

    def end_all():
        for q in procs:
            if q.poll() is None:
                q.send_signal(signal.SIGTERM)
        t_end = time.time() + 10.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    t0, rc = time.time(), 0
    while True:
        codes = [q.poll() for q in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            sys.stderr.write("bench: rank %d exited with code %d; ending the other ranks\n" % bad[0])
            rc = bad[0][1] if bad[0][1] > 0 else 1
            grace = time.time() + 5.0                       # a clean collective failure brings the others down by itself
            while time.time() < grace and any(q.poll() is None for q in procs):
                time.sleep(0.05)
            end_all()
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > limit:
            sys.stderr.write("bench: the %d ranks did not finish within %.0f s\n" % (n, limit))
            end_all()
            rc = 124
            break
        time.sleep(0.05)
    reader.join(timeout=5.0)
    lines = [ln for ln in box["out"].decode(errors="replace").splitlines() if ln.strip()]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench: rank 0 printed no result line\n")
        rc = 5
    return rc


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="A", choices=["A", "C", "E"])
    ap.add_argument("--path", default="fused", choices=["fused", "ops", "opsgraph"])
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU for workload A (default 128; N>1: 1024/N)")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="N>1, workload A: which curve `value` is on (the other one is reported beside it)")
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--min-ms", type=float, default=50.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary objects (config_C, paths, ...)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--epoch-loop-only", action="store_true", help="N = 1: print just the epoch_loop object")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` with no launcher: become the launcher BEFORE anything touches the GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])

    # stdout carries exactly ONE line, the JSON result: native libraries print there too (RCCL writes a version /
    # hostname banner to stdout when a communicator is created), so file descriptor 1 is pointed at stderr for the
    # whole run and the result goes to a saved duplicate of the original stdout — once, whoever gets there first.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    emit_lock, emitted = threading.Lock(), [False]

    def emit(obj):
        with emit_lock:
            if emitted[0] or obj is None:
                return
            emitted[0] = True
            os.write(result_fd, (json.dumps(obj) + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        args.gpus = world                                  # the launcher's WORLD_SIZE is authoritative

    # ORDER MATTERS: torch first, libtnn_hip.so second (one HIP runtime per process).  torch itself is
    # only the control plane (gloo rendezvous / barrier) and the contract's torch.cuda.synchronize().
    import torch
    import torch.distributed                              # noqa: F401
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("TNN_DEVICE", local_rank)))   # TNN_DEVICE: ranks sharing one GPU (tests)
    lib = _lib.get()                                    # binds LOCAL_RANK's GPU; raises without HIP
    assert tn.backend_name() == "hip-gfx950", "bench.py measures the HIP library only"
    comm = tn.dist.init_from_env() if (world > 1 or os.environ.get("TNN_FORCE_COMM") == "1") else None
    force_dp = comm is not None and world == 1
    if os.environ.get("TNN_BENCH_TEST_EXIT_RANK") == str(rank) and world > 1:
        os._exit(9)                                      # test hook: this rank dies after the rendezvous (tests/test_gpu_p2p.py)
    clock = Clock(torch, comm, world)
    solo = Clock(torch, None, 1)                         # rank-local measurements (no barrier, no max over ranks)
    use_graph = not args.no_graph
    line = None
    exit_code = 0

    if args.epoch_loop_only:
        emit({"epoch_loop": epoch_loop_object(float(os.environ.get("TNN_HEADLINE", "5.98e6")))})
        return 0
    if args.workload == "A":
        widths, kind = WIDTHS_A, "softmax_nll"
        if args.rows is not None:
            rows = args.rows
        elif world > 1 and args.scaling == "strong":
            if GLOBAL_BATCH_D % world:
                raise SystemExit("strong scaling splits the global batch of %d evenly: %d ranks do not" % (GLOBAL_BATCH_D, world))
            rows = GLOBAL_BATCH_D // world
        else:
            rows = 128
        steps = args.steps if args.steps is not None else 2000
        warmup = args.warmup if args.warmup is not None else 64
        n_batches = 64 if rows <= 256 else 32
    elif args.workload == "C":
        widths, rows, kind = WIDTHS_C, args.rows or 512, "mse"
        steps = args.steps if args.steps is not None else 50
        warmup = args.warmup if args.warmup is not None else 5
        n_batches = 2
    else:
        widths, rows, kind = WIDTHS_E, args.rows or 512, "mse"
        steps = args.steps if args.steps is not None else 20
        warmup = args.warmup if args.warmup is not None else 3
        n_batches = 2

    # ---------------------------------------------------------------- primary measurement
    transports = None
    if args.workload == "E":
        args.path = "fused"
        runner = FusedRun(widths, rows, kind, n_batches, rank, world, comm, force_dp, dtype="bfloat16")
    elif args.path == "fused":
        runner = FusedRun(widths, rows, kind, n_batches, rank, world, comm, force_dp, use_graph=use_graph)
    else:
        runner = OpsRun(widths, rows, kind, min(n_batches, 16), rank, world, comm, graph=args.path == "opsgraph")

    def replicas_identical(r):
        crc = r.params_crc()
        if world > 1:
            import torch.distributed as dist
            box = [None] * world
            dist.all_gather_object(box, crc)
            return bool(all(c == box[0] for c in box))
        return True

    def all_ranks(flag):
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([1 if flag else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))
        return bool(flag)

    def p2p_alive():
        st = comm.p2p_status() if comm is not None and hasattr(comm, "p2p_status") else None
        return all_ranks(bool(st and st["enabled"] and not st["dead"]))

    selftest_before = None
    if comm is not None and args.path == "fused":
        selftest_before = collective_selftest(comm, world, rank, all_ranks)
    if comm is not None and args.path == "fused" and isinstance(runner, FusedRun):
        # Data-parallel run.  RCCL first (north_star's named transport), the xGMI peer-to-peer path second; both are
        # reported.  The latency path carries f32 sums up to its mapped capacity: config A's 0.94 MB arena, not C's
        # 134 MB or E's 1 GB — those go to RCCL whatever the transport's state.
        transports = {}
        arena_bytes = (int(runner.trainer.arena_size) + 1) * 4
        have_rccl = bool(getattr(comm, "_rccl", False))
        have_p2p = p2p_alive() and arena_bytes <= getattr(comm, "p2p_bytes", 0)
        res_rccl = res_p2p = None
        if have_rccl:
            if have_p2p:
                comm.set_p2p(False)
                runner.capture()
            res_rccl = measure(clock, runner, warmup, steps, args.repeats, args.min_ms, rows * world)
            transports["rccl"] = brief(res_rccl, replicas_identical=replicas_identical(runner),
                                       graph_captured=runner.chunk is not None)
        primary, used = res_rccl, "rccl"
        if have_p2p:
            # nothing below may cost the RCCL result: a watchdog emits the line as it stands (RCCL as `value`) and ends
            # the process with a NON-ZERO code if the peer-to-peer run does not come back (bounded spins make that a
            # 20 s affair per stuck barrier; a hard hang is what the timer is for)
            limit = int(os.environ.get("TNN_BENCH_P2P_TIMEOUT_S", "120"))
            partial = {"line": None}

            def give_up():
                if partial["line"] is not None:
                    partial["line"]["exit_code"] = 3
                    partial["line"]["config"]["collectives"]["xgmi_p2p"] = "did not finish in %d s" % limit
                    emit(partial["line"])
                os._exit(3)
            dog = threading.Timer(limit, give_up)
            dog.daemon = True
            if res_rccl is not None and rank == 0:
                partial["line"] = make_line(args, widths, rows, kind, world, warmup, steps, res_rccl, runner,
                                            dict(transports, used="rccl"), force_dp)
            if res_rccl is not None:
                dog.start()
            comm.set_p2p(True)
            try:
                runner.capture()
                res_p2p = measure(clock, runner, warmup, steps, args.repeats, args.min_ms, rows * world)
            except Exception as exc:                              # noqa: BLE001 - a peer timeout raises on every rank (comm.check votes)
                sys.stderr.write("bench: the peer-to-peer leg raised: %s\n" % exc)
                res_p2p = None
            if res_p2p is None:
                dog.cancel()
                transports["xgmi_p2p"] = "failed (a peer barrier timed out); transport switched off"
                if res_rccl is None:
                    raise SystemExit("bench: the peer-to-peer transport failed and no RCCL communicator exists")
                comm.set_p2p(False)
                runner.capture()
            else:
                alive = p2p_alive()
                try:
                    verified = all_ranks(alive and comm.p2p_selftest(sizes=(235147, 4099, 65536), rounds=6))
                except Exception as exc:                          # noqa: BLE001 - every rank must reach the vote
                    sys.stderr.write("bench: post-run peer-to-peer check raised: %s\n" % exc)
                    verified = all_ranks(False)
                same = replicas_identical(runner)
                dog.cancel()
                transports["xgmi_p2p"] = brief(res_p2p, barrier_timed_out=not alive, verified_after_run=verified,
                                               replicas_identical=same, graph_captured=runner.chunk is not None)
                if verified and same and alive:
                    primary, used = res_p2p, "xgmi-p2p"
                elif res_rccl is None:
                    raise SystemExit("bench: the peer-to-peer transport failed its checks and no RCCL communicator exists")
        transports["used"] = used
        transports["rule"] = "value = xgmi_p2p when verified bit-exact after the run, no barrier timed out and replicas identical; else rccl"
        if primary is None:
            raise SystemExit("bench: no usable transport")
        res = primary
    else:
        res = measure(clock, runner, warmup, steps, args.repeats, args.min_ms, rows * world)

    if comm is not None and transports is not None:
        comm.set_p2p(transports["used"] == "xgmi-p2p")       # everything below runs on the primary transport
    check = check_timed = None
    if args.path == "fused" and args.workload == "A":
        # the reference's fixtures exist at global batches 128 and 1024: a weak-scaling line at N = 2 / 4 (global batch 256 /
        # 512) checks the same trainer + transport at config D's split instead (1024 / N rows per rank, the step form its
        # strong_scaling point times)
        rows_chk = rows
        if world > 1 and rows * world not in (128, GLOBAL_BATCH_D) and GLOBAL_BATCH_D % world == 0:
            rows_chk = GLOBAL_BATCH_D // world
        check = fixture_check(widths, rows_chk, kind, rank, world, comm, force_dp, use_graph)
        if check is not None:
            check["ok"] = all_ranks(check["ok"])
            check["rows_per_rank"], check["global_batch"] = rows_chk, rows_chk * world
            check["step_form"] = ("single-GPU 2L - 2 launch step" if comm is None else
                                  "merged 2L - 2 launch data-parallel step, %d block(s) of <= 128 rows per rank" % ((rows_chk + 127) // 128))
        if comm is not None and not args.no_extras:
            # ... and the step form actually TIMED (its own rows per rank), against the single-GPU trainer on the whole batch
            check_timed = timed_rows_check(widths, rows, kind, rank, world, comm, force_dp, use_graph, torch)
            ok_t = all_ranks(check_timed["ok"] if check_timed is not None else True)
            if check_timed is not None:
                check_timed["ok"] = ok_t
    if rank == 0:
        line = make_line(args, widths, rows, kind, world, warmup, steps, res, runner, transports, force_dp)
        if check is not None:
            line["parity_vs_reference_fixture"] = check
            if not check["ok"]:
                exit_code = line["exit_code"] = 4            # a fast step with the wrong losses is not a result
        if args.path == "fused" and args.workload == "A" and check_timed is not None:
            line["parity_at_timed_rows"] = check_timed
            if not check_timed["ok"]:
                exit_code = line["exit_code"] = 4

    # ---------------------------------------------------------------- scaling curves (workload A)
    if args.workload == "A" and args.path == "fused" and not args.no_extras and args.rows is None:
        other_rows = None
        if world > 1:
            other_rows = 128 if args.scaling == "strong" else GLOBAL_BATCH_D // world
        point = {"global_batch": rows * world, "rows_per_rank": rows, "value": round(res["value"], 1),
                 "ms_per_step": round(res["ms_per_step"], 5)}
        curves = {("strong_scaling" if (world > 1 and args.scaling == "strong") else "weak_scaling"): point}
        strong_note = ("strong scaling of configs[3] (global batch 1024 split over N ranks) is bounded by launch latency, not by "
                       "work: the per-rank step costs about the same number of dependent launches whatever its row count, so the "
                       "ceiling at N ranks is (single-GPU bs-1024 step) / (bs-1024/N sharded step incl. two collectives); see "
                       "DESIGN.md §7 for the measured per-row-count steps.  The weak curve (128 rows per rank) is reported beside it.")
        if world == 1:
            # N = 1 point of the strong curve: the whole global batch of config D on one GPU
            d1 = FusedRun(widths, GLOBAL_BATCH_D, kind, 32, 0, 1, None, False, use_graph=use_graph)
            r1 = measure(solo, d1, warmup, steps, 3, args.min_ms, GLOBAL_BATCH_D)
            curves["strong_scaling"] = brief(r1, global_batch=GLOBAL_BATCH_D, rows_per_rank=GLOBAL_BATCH_D,
                                             launches_per_step=d1.launches_per_step(), note=strong_note)
            del d1
            if not args.no_extras and args.rows is None:
                # the same net at the batch sizes in between (the per-rank batches of the strong curve at N = 4 / 2)
                between = {}
                for rows_b in (256, 512):
                    rb_run = FusedRun(widths, rows_b, kind, 32, 0, 1, None, False, use_graph=use_graph)
                    between[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b),
                                                 launches_per_step=rb_run.launches_per_step())
                    del rb_run
                curves["batch_sizes"] = between
                # the reference's OWN example net (examples/mnist/run.py:59-69), same batch size and optimizer: the trainer's
                # 2 L - 2 = 8 launch step (hidden widths padded to multiples of 16, generic merged head kernel); pinned against the reference
                # by tests/golden/traj_R_example.npz
                ex_widths = [784, 200, 100, 70, 30, 10]
                ex_run = FusedRun(ex_widths, 128, kind, 64, 0, 1, None, False, use_graph=use_graph)
                curves["reference_example_net"] = brief(measure(solo, ex_run, warmup, steps, 3, args.min_ms, 128),
                                                        widths="-".join(map(str, ex_widths)), rows=128,
                                                        launches_per_step=ex_run.launches_per_step())
                del ex_run
                ex_sizes = {}
                for rows_b in (256, 512, 1024):              # the generic merged head walking 2 / 4 / 8 blocks of 128 rows
                    rb_run = FusedRun(ex_widths, rows_b, kind, 32, 0, 1, None, False, use_graph=use_graph)
                    ex_sizes[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b),
                                                  launches_per_step=rb_run.launches_per_step())
                    del rb_run
                curves["reference_example_net"]["batch_sizes"] = ex_sizes
            curves["weak_scaling_1024"] = dict(curves["strong_scaling"], note="N = 1 point of the third curve (1024 rows per rank): "
                                               "the same measurement as strong_scaling's N = 1 point")
        elif other_rows != rows:
            other = FusedRun(widths, other_rows, kind, 64 if other_rows <= 256 else 32, rank, world, comm, False,
                             use_graph=use_graph)
            ro = measure(clock, other, warmup, steps, 3, args.min_ms, other_rows * world)
            curves["weak_scaling" if args.scaling == "strong" else "strong_scaling"] = brief(
                ro, global_batch=other_rows * world, rows_per_rank=other_rows, transport=transports["used"],
                replicas_identical=replicas_identical(other))
            del other
        if world > 1:
            # third curve: 1024 rows per rank (global batch 1024 N) — the definition under which the step is long enough for the
            # exchange to amortise (ceilings of the three curves)
            w1024 = FusedRun(widths, 1024, kind, 32, rank, world, comm, False, use_graph=use_graph)
            rw = measure(clock, w1024, warmup, steps, 3, args.min_ms, 1024 * world)
            curves["weak_scaling_1024"] = brief(rw, global_batch=1024 * world, rows_per_rank=1024, transport=transports["used"],
                                                replicas_identical=replicas_identical(w1024))
            del w1024
            # the single-GPU references of ALL curves, measured in THIS run on rank 0 while the others wait; each curve's
            # speedup is computed on its own definition (weak: 128 rows on one GPU; strong: the whole 1024 rows on one GPU)
            if rank == 0:
                for rows_1 in (128, GLOBAL_BATCH_D):
                    d1 = FusedRun(widths, rows_1, kind, 64 if rows_1 <= 256 else 32, 0, 1, None, False, use_graph=use_graph)
                    r1 = measure(solo, d1, warmup, steps, 3, args.min_ms, rows_1)
                    curves["single_gpu_bs%d" % rows_1] = brief(r1, note="rank 0 alone, no communicator")
                    del d1
                for name, rows_1 in (("weak_scaling", 128), ("strong_scaling", GLOBAL_BATCH_D), ("weak_scaling_1024", 1024)):
                    if name in curves:
                        curves[name]["speedup_vs_n1"] = round(
                            curves[name]["value"] / curves["single_gpu_bs%d" % rows_1]["value"], 4)
                own = "strong_scaling" if args.scaling == "strong" else "weak_scaling"
                if line is not None:
                    line["speedup_vs_n1"] = curves[own]["speedup_vs_n1"]
            comm.barrier()
        if "strong_scaling" in curves:
            curves["strong_scaling"].setdefault("note", strong_note)
        if line is not None:
            line.update(curves)

    # ---------------------------------------------------------------- what the communicator ran on (every line with one)
    if comm is not None and args.path == "fused" and not args.no_extras:
        after = collective_selftest(comm, world, rank, all_ranks)
        table = collective_latency_table(comm, clock)
        if line is not None:
            used = transports["used"] if transports else None
            line["multi_gpu"] = {
                "world": world, "rccl_version": rccl_version_string(), "topology": topology_object(torch),
                "selftest_before_timed_runs": selftest_before, "selftest_after_timed_runs": after,
                "collective_latency": table,
                "value_from": used,
                "why": (transports or {}).get("rule"),
                "ranks_share_a_device": os.environ.get("TNN_DEVICE") is not None,
            }
        if transports is not None:
            comm.set_p2p(transports["used"] == "xgmi-p2p")

    # ---------------------------------------------------------------- forced communicator at world 1: the step forms of N > 1
    if comm is not None and world == 1 and args.workload == "A" and args.path == "fused" and not args.no_extras and args.rows is None:
        dp_sizes = {}
        for rows_b in (256, 512, 1024):                      # the per-rank batches of the strong curve and of weak_scaling_1024
            rb_run = FusedRun(widths, rows_b, kind, 32, 0, 1, comm, True, use_graph=use_graph)
            dp_sizes[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b),
                                          graph_captured=rb_run.chunk is not None)
            del rb_run
        ex_widths = [784, 200, 100, 70, 30, 10]
        ex_dp = {}
        for rows_b in (128, 1024):
            rb_run = FusedRun(ex_widths, rows_b, kind, 32, 0, 1, comm, True, use_graph=use_graph)
            ex_dp[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b),
                                       graph_captured=rb_run.chunk is not None)
            del rb_run
        if line is not None:
            line["dp_world1_batch_sizes"] = dict(dp_sizes, note="the data-parallel step (both collectives issued, world 1) at the "
                                                 "per-rank batches of the strong curve (256 / 512) and of weak_scaling_1024; transport: %s"
                                                 % (transports["used"] if transports else "rccl"))
            line.setdefault("reference_example_net", {"widths": "-".join(map(str, ex_widths))})["dp_world1"] = ex_dp

    # ---------------------------------------------------------------- configs[4] (bf16, 8 GPUs) on the data-parallel line
    if (comm is not None and (getattr(comm, "_rccl", False) or getattr(comm, "p2p_bulk_bytes", 0) > 0) and args.workload == "A" and args.path == "fused"
            and not args.no_extras and args.rows is None and os.environ.get("TNN_BENCH_CONFIG_E", "1") != "0"):
        # never at the price of the line: a watchdog on EVERY rank emits the line as it stands and ends the process if the
        # extra measurement does not come back (it is the first time this step form meets real links)
        limit_e = int(os.environ.get("TNN_BENCH_CONFIG_E_TIMEOUT_S", "180"))
        state_e = {"note": "did not finish in %d s" % limit_e}

        def stop_e():
            # a hung sharded-optimizer measurement is NOT a successful run: the line is emitted as it stands, marked, and the
            # process ends with a code of its own (6) so that self_launch / the driver see the failure
            if line is not None:
                line["config_E"] = state_e["note"]
                line["exit_code"] = line.get("exit_code") or 6
            emit(line)
            os._exit(exit_code or 6)
        dog_e = threading.Timer(limit_e, stop_e)
        dog_e.daemon = True
        dog_e.start()
        if getattr(comm, "_rccl", False):
            comm.set_p2p(False)                              # bandwidth-sized messages: RCCL
        # (a peer-to-peer-only group — TNN_COMM=xgmi, ranks sharing a GPU — carries them on the transport's bulk path)
        try:
            obj_e = config_e_object(clock, rank, world, comm, force_dp)
        except Exception as exc:                             # noqa: BLE001
            # the other ranks may be inside a collective of the measurement: no vote is possible — wait for the watchdogs
            state_e["note"] = "failed on rank %d: %s" % (rank, exc)
            sys.stderr.write("bench: config_E %s\n" % state_e["note"])
            time.sleep(limit_e + 30)
            obj_e = state_e["note"]
        dog_e.cancel()
        if transports is not None:
            comm.set_p2p(transports["used"] == "xgmi-p2p")
        if line is not None:
            line["config_E"] = obj_e

    # ---------------------------------------------------------------- secondary objects (rank 0, N = 1)
    if line is not None and world == 1 and not args.no_extras:
        if args.workload == "E":
            line["roofline"] = dw_adam_roofline_in_step(runner, widths, rows, res["ms_per_step"])
            line["gemm_roofline"] = time_gemms_bf16(widths, rows)
        elif args.workload == "C":
            line["roofline"] = time_gemms(widths, rows, reps=20)
            attach_gemm_traffic(line["roofline"], "C")
        else:
            line["roofline"] = latency_roofline(widths, rows, res, runner)
            line["roofline_gemm4096"] = None               # (key order of the line; measured below, BEHIND the epoch loop)
            if args.path == "fused" and args.rows is None and comm is None:
                paths = {}
                for name, graph in (("ops_eager", False), ("ops_graph", True)):
                    r = OpsRun(widths, rows, kind, 16, graph=graph)
                    paths[name] = brief(measure(solo, r, 20, 200, 3, args.min_ms, rows))
                    del r
                paths["note"] = ("the same step on the drop-in Tensor/ops/Dense/ReLU/SoftmaxCrossEntropyLoss/Adam/Model API (the "
                                 "reference's loop body, 4 launches per step like the trainer): issued from Python op by op (eager) / "
                                 "recorded with tn.capture, 16 steps on their resident batches per hipGraph, and replayed (graph)")
                paths["host_modules"] = ("compiled ahead of time from the .py sources (tinynn-autograd_amd/_host_build.py)"
                                         if tn.host_modules_compiled() else "interpreted")
                paths["host_call_wrappers"] = ("%d of %d entry points called through generated C wrappers instead of ctypes "
                                               "(tinynn-autograd_amd/_fastcall_gen.py)" % (_lib.get().fast_calls, len(_lib._SIGNATURES)))
                line["paths"] = paths
                # (before the large configurations: releasing their GBs of buffers stalls the GPU once, ~70 ms, some 50 ms later —
                # tools/probes/epoch_stall.py; the object's `value` is a median over the steady epochs anyway)
                # (round 6: also before the 4096-wide GEMM replays — a sustained power-limited load in front is what brings the
                # one-off 35-80 ms GPU-side pause into one of the epochs, profiles/r06_epoch_stall_clocks.txt)
                line["epoch_loop"] = epoch_loop_object(res["value"])
                line["roofline_gemm4096"] = time_gemms(WIDTHS_C, 512, reps=20)
                attach_gemm_traffic(line["roofline_gemm4096"], "C")
                c = FusedRun(WIDTHS_C, 512, "mse", 2, use_graph=use_graph)
                rc = measure(solo, c, 3, 20, 3, 0.0, 512)
                line["config_C"] = brief(rc, workload="configs[2]: Dense 4096-4096-4096 autoencoder, bs 512, sum-of-squares/m, Adam",
                                         algorithmic_gflop_per_step=85.8993,
                                         mfma_frac_of_whole_step=round(85.8993e9 / (rc["ms_per_step"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4))
                del c
                line["config_E"] = config_e_object(solo)
            if line.get("roofline_gemm4096") is None:
                line["roofline_gemm4096"] = time_gemms(WIDTHS_C, 512, reps=20)
                attach_gemm_traffic(line["roofline_gemm4096"], "C")
        if not args.no_cpu_baseline and args.workload != "E":
            line["cpu_baseline"] = cpu_baseline(widths, rows, kind, budget_s=8.0 if args.workload == "A" else 15.0)

    if (rank == 0 and world == 1 and comm is None and line is not None and not args.no_extras and args.workload == "A"
            and args.path == "fused" and args.rows is None):
        # the data-parallel step forms with a one-rank communicator (every collective issued; what N > 1 runs per rank), AFTER
        # every other measurement of this line
        os.environ["TNN_FORCE_COMM"] = "1"
        comm1 = None
        try:
            comm1 = tn.dist.init_from_env()
            used1 = "xgmi-p2p" if getattr(comm1, "_p2p", False) else "rccl"
            dp1 = {"transport": used1, "note": "one-rank communicator, both collectives issued; rows per rank as on the N > 1 curves"}
            for rows_b in (128, 1024):
                rb_run = FusedRun(widths, rows_b, kind, 32, 0, 1, comm1, True, use_graph=use_graph)
                dp1[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b), graph_captured=rb_run.chunk is not None)
                del rb_run
            line["dp_world1"] = dp1
            ex_widths = [784, 200, 100, 70, 30, 10]
            ex_dp = {"transport": used1}
            for rows_b in (128, 1024):
                rb_run = FusedRun(ex_widths, rows_b, kind, 32, 0, 1, comm1, True, use_graph=use_graph)
                ex_dp[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b), graph_captured=rb_run.chunk is not None)
                del rb_run
            line.setdefault("reference_example_net", {})["dp_world1"] = ex_dp
        except Exception as exc:                             # noqa: BLE001 - an extra object never costs the line
            line["dp_world1"] = "unavailable: %s: %s" % (type(exc).__name__, exc)
        finally:
            os.environ.pop("TNN_FORCE_COMM", None)
            if comm1 is not None and hasattr(comm1, "close"):
                comm1.close()
    if rank == 0 and world == 1 and line is not None and not args.no_extras:
        line["box"] = box_object(line)
    emit(line)
    if comm is not None:
        comm.barrier()
        if hasattr(comm, "close"):
            comm.close()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([exit_code])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exit_code = int(t.item())
    return exit_code


def box_probe():
    """tools/probes/bin/libtnn_probe.so (its own library: `make -C tinynn-autograd_amd/csrc probe`, built by
    __graft_entry__.build()) -> what this box's MFMA pipes, clocks and HBM do right now (~100 ms on the GPU)."""
    path = os.path.join(ROOT, "tools", "probes", "bin", "libtnn_probe.so")
    lib = ctypes.CDLL(path)
    lib.tnn_probe_box.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    lib.tnn_probe_box.restype = ctypes.c_int
    out = (ctypes.c_double * 10)()
    rc = lib.tnn_probe_box(out, 10)
    if rc:
        raise RuntimeError("tnn_probe_box failed with code %d" % rc)
    return {"mfma_f32_tflops": round(out[0], 1), "mfma_f32_clock_ghz": round(out[1], 3),
            "mfma_f32_tflops_step_like_operands": round(out[8], 1), "mfma_f32_clock_ghz_step_like_operands": round(out[9], 3),
            "mfma_bf16_tflops_random_operands": round(out[2], 1), "mfma_bf16_clock_ghz_random_operands": round(out[3], 3),
            "mfma_bf16_tflops_zero_operands": round(out[4], 1), "mfma_bf16_clock_ghz_zero_operands": round(out[5], 3),
            "copy_float4_gbs": round(out[6], 1), "stream_4read_3write_gbs": round(out[7], 1)}


def box_object(line):
    """What THIS box can do (libtnn_probe.so: MFMA-only loops with random / zero operands and their sustained clocks, a float4
    copy over 2 GiB), measured after everything else so that it does not disturb the timed runs — and every MFMA- or
    HBM-bound roofline object on the line gets `frac_of_box` beside its spec-peak `frac`: achieved / the same box's probe
    (bf16 against the random-operand loop: the chip clocks to its power budget and real data is not zeros)."""
    box = box_probe()
    box["note"] = ("MFMA-only loops: 8 waves per CU, 8 independent accumulators; spec peaks 157.3 (fp32) / 2500 (bf16 dense) "
                   "TFLOP/s, 8000 GB/s; frac_of_box on the roofline objects = achieved / this box's probe")

    def annotate(obj):
        if isinstance(obj, dict):
            if obj.get("bound") in ("mfma", "hbm") and isinstance(obj.get("achieved"), (int, float)):
                if obj["bound"] == "hbm":
                    ref = max(box["copy_float4_gbs"], box["stream_4read_3write_gbs"])
                else:
                    ref = box["mfma_bf16_tflops_random_operands"] if obj.get("peak") == PEAK_BF16_MFMA_TFLOPS else box["mfma_f32_tflops"]
                if ref:
                    obj["box_peak"] = ref
                    obj["frac_of_box"] = round(obj["achieved"] / ref, 4)
            for v in obj.values():
                annotate(v)
        elif isinstance(obj, list):
            for v in obj:
                annotate(v)
    annotate(line)
    return box


def latency_roofline(widths, rows, res, runner):
    """Config A/D: neither MFMA nor HBM bounds the step — the launch chain does."""
    flops, gemm_bytes, adam_bytes = step_algorithmic(widths, rows)
    step_us = res["ms_per_step"] * 1e3
    roof = {"bound": "latency", "unit": "ksteps/s", "achieved": round(1e3 / step_us, 3)}
    gem = time_gemms(widths, rows, reps=200)
    if isinstance(runner, FusedRun) and runner.comm is None:
        launches = runner.launches_per_step()
        per = runner.per_launch_us()
        floor_us = launches * LAUNCH_BOUNDARY_US
        roof.update({"peak": round(1e3 / floor_us, 3), "frac": round(floor_us / step_us, 4),
                     "launches_per_step": launches, "launch_boundary_us": LAUNCH_BOUNDARY_US,
                     "launch_floor_us_per_step": round(floor_us, 3), "step_us": round(step_us, 3),
                     "per_launch_us": per, "sum_per_launch_us": round(sum(per), 3),
                     "model": "peak = 1 / (launches x dependent-kernel boundary); each per_launch_us is that launch replayed "
                              "back to back (HIP events), i.e. boundary + kernel"})
    else:
        roof.update({"peak": None, "frac": None, "step_us": round(step_us, 3)})
    traffic, src = step_traffic("A")
    roof["traffic"] = traffic
    if src:
        roof["traffic_unit"] = "HBM-side bytes per step, all kernels of the step (PMC, %s)" % src
    if "per_launch_us" in roof and len(widths) >= 3:
        # the step's largest launch against the roofline that would bound it if anything but latency did: the first layer's
        # backward with the whole optimizer step in it (HBM: x, dz0, every parameter's p / m / v read and written)
        n_params = sum(widths[l] * widths[l + 1] + widths[l + 1] for l in range(len(widths) - 1))
        rest = n_params - (widths[0] * widths[1] + widths[1])
        alg = 4 * (rows * widths[0] + rows * widths[1]) + 24 * n_params + 4 * rest
        us = roof["per_launch_us"][-1]
        dom = {"kernel": "dense_bwd0_adam_kernel<4> (dW0 = x^T dz0 + db0 with Adam over the whole parameter arena in the launch)",
               "bound": "hbm", "achieved": round(alg / us / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": round(alg / us / 1e3 / PEAK_HBM_GBS, 4), "traffic": None, "algorithmic_bytes": int(alg), "us": round(us, 3),
               "us_note": "HIP events over back-to-back replays of this launch alone (boundary + kernel)",
               "model": "x [rows, n_in] and dz0 [rows, n_1] read; p, m, v of every parameter read and written (24 B/param); the other "
                        "layers' gradients read (4 B/param); dW0 itself never stored"}
        table, tsrc = load_traffic_table()
        if table is not None:
            for name, per in table["kernels"].items():
                if "A" in per and name.startswith("dense_bwd0_adam_kernel<"):
                    dom["traffic"] = per["A"]["fetch_bytes"] + per["A"]["write_bytes"]
                    dom["traffic_unit"] = "HBM-side bytes per launch (PMC: FETCH_SIZE x 2 + WRITE_SIZE, %s)" % tsrc
        roof["dominant_kernel"] = dom
    roof.update({"algorithmic_bytes": int(gemm_bytes + adam_bytes), "algorithmic_gflop_per_step": round(flops / 1e9, 4),
                 "hbm_frac": round((gemm_bytes + adam_bytes) / (step_us * 1e-6) / (PEAK_HBM_TBS * 1e12), 4),
                 "mfma_frac_of_whole_step": round(flops / (step_us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                 "gemm_frac": gem["frac"], "gemm_tflops": gem["achieved"], "gemm_us_per_step": gem["gemm_us_per_step"],
                 "per_gemm": gem["per_gemm"]})
    return roof


def make_line(args, widths, rows, kind, world, warmup, steps, res, runner, transports, force_dp):
    cfg_a = "configs[1]" if world == 1 and rows == 128 else "configs[3]"
    if world > 1 and rows == 128:
        cfg_a = "configs[1] per rank, data-parallel over %d ranks%s" % (world, " = configs[3]" if world * rows == GLOBAL_BATCH_D else "")
    cfg_name = {"A": cfg_a, "C": "configs[2]", "E": "configs[4]"}[args.workload]
    graph = getattr(runner, "chunk", None) is not None
    scaling = "weak"
    if args.workload == "A" and world > 1 and args.rows is None:
        scaling = args.scaling
    line = {
        "metric": {"A": "training samples/sec, MNIST 3-layer MLP (784-256-128-10), bs=128, at 1/2/4/8 GPUs",
                   "C": "training samples/sec, Dense 4096-4096-4096 autoencoder, bs=512",
                   "E": "training samples/sec, 8192-wide 4-layer MLP bf16, bs=512 per GPU"}[args.workload],
        "value": round(res["value"], 1), "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(res["ms_per_step"], 5), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "bf16 (fp32 accumulate, fp32 master weights)" if args.workload == "E" else "f32",
        "data": "synthetic",
        "timing": {"statistic": "median of %d repeats of [%d warm-up steps + %d timed steps] (max over ranks each)"
                                % (res["repeats"], warmup, res["timed_steps_per_repeat"]),
                   "min_ms_per_step": round(res["min_ms_per_step"], 5), "max_ms_per_step": round(res["max_ms_per_step"], 5),
                   "segments_per_repeat": res["segments_per_repeat"], "timed_ms_per_repeat": round(res["ms_per_step"] * res["timed_steps_per_repeat"], 2)},
        "config": {"workload": "%s: Dense/ReLU MLP %s, %d rows per GPU (global batch %d), whole-batch "
                               "softmax NLL%s, Adam lr=1e-3" % (
                                   cfg_name, "-".join(map(str, widths)), rows, rows * world,
                                   "" if kind == "softmax_nll" else " replaced by sum-of-squares/m"),
                   "path": args.path + ("+hipGraph(%d steps/launch)" % runner.n_batches if graph else "")
                           + ("+comm(world=1, forced)" if force_dp else ""),
                   "parallelism": "dp%d" % world, "global_batch": rows * world, "rows_per_rank": rows,
                   "data_resident_in_hbm": True,
                   **({"collectives": transports} if transports else {})},
        "final_loss": round(res["final_loss"], 6),
        "device": _lib.device_props()["name"],
        "exit_code": 0,
    }
    return line


if __name__ == "__main__":
    sys.exit(main())
