"""The bench contract's timed region (barrier + stream sync + torch.cuda.synchronize() on both sides, max over ranks) and the
median-of-repeats protocol around it."""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

from .common import _lib    # noqa: F401


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
