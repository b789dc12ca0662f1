"""Objects of the JSON line that are whole measurements of their own (config E, the epoch loop) and the line itself."""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

from .common import GLOBAL_BATCH_D, PEAK_BF16_MFMA_TFLOPS, WIDTHS_E, _lib    # noqa: F401
from .common import brief
from .clock import measure
from .runners import FusedRun
from .roofline import dw_adam_roofline_in_step, time_gemms_bf16


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


def all_epochs_object(stats, steady, num_ep, n_train, train_all):
    """Everything from the first shuffle to the last loss — and the same WITHOUT any paused epoch (one whose `steps` time is more
    than 3x the steady median).  The one-off 35-80 ms epoch of rounds 4-6 was the container's CPU-bandwidth controller stopping
    every thread of the process: numpy's 64-thread BLAS pool spinning behind the dataset's matmul spent the 16-CPU quota
    (profiles/r06_epoch_stall_root_cause.txt).  bench.py now sizes the pool to the quota at start (utils/host_threads.py), so
    `paused_epochs` is expected to be empty; the detection stays as a tripwire."""
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
                   "capture of the epoch graph (phases_per_epoch.capture[0]); epochs >= 1 replay it (capture 0.0).  `value` is the "
                   "median of the steady epochs and every epoch is listed.  (The one-off 50-80 ms epoch earlier rounds reported was the "
                   "container's CPU quota throttling the process behind a 64-thread BLAS pool, not the GPU: "
                   "profiles/r06_epoch_stall_root_cause.txt; the pool is now sized to the quota at start.)")
    return out


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
