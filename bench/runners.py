"""What is timed: runners that replay training steps from hipGraphs (whole-step trainer, op-level API), and the parity checks of
the very trainer + transport being timed against the reference fixtures."""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

from .common import ROOT, WIDTHS_A, _lib, da, tn    # noqa: F401
from .common import build_net, events_us, synth_batches


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
