"""Shared by every module of the bench package: the library imports, the peaks and widths BASELINE.json / the MI355X guide name,
synthetic inputs, the net builder, the GEMM list of a step, HIP-event timing of a replayed launch, result formatting."""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # the repository root (this package sits in it)
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


def brief(res, **extra):
    out = {"value": round(res["value"], 1), "unit": "samples/s", "ms_per_step": round(res["ms_per_step"], 5),
           "min_ms_per_step": round(res["min_ms_per_step"], 5), "max_ms_per_step": round(res["max_ms_per_step"], 5),
           "final_loss": round(res["final_loss"], 6)}
    out.update(extra)
    return out
