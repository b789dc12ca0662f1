"""Data-parallel plumbing: one process per GPU, RCCL over xGMI for the data plane.

New relative to the reference (no communication there).  Two collectives exist on the path:
  C1  in-place SUM all-reduce of the flat gradient arena between backward() and the optimizer step;
  C2  an all-gather of one {max, sum-exp} pair per rank inside the whole-batch softmax loss
      (core/losses.py:26-27 couples the shards), merged with a log-sum-exp kernel.

`torch.distributed` (gloo) is used only as the control plane: rendezvous from the RANK / WORLD_SIZE /
MASTER_* environment that `python -m torch.distributed.run` provides, broadcast of the RCCL unique id,
barriers and the max-over-ranks of bench timings.  Import torch BEFORE the first device call of this package
(single HIP runtime per process, see init_from_env).  `GlooCommunicator` moves the same two collectives over
gloo through host memory; it exists for the world_size-2 CPU tests.
"""

import atexit
import ctypes
import os
import sys

import numpy as np

from . import _lib
from . import device_array as da


class Communicator(object):
    rank = 0
    world = 1

    def allreduce(self, arr, op="sum"):
        raise NotImplementedError

    def allgather(self, arr):
        """[n] per rank -> [world, n]"""
        raise NotImplementedError

    def barrier(self):
        pass

    def merge_softmax_stats(self, stats):
        """{M_r, S_r} of every shard -> global {M, S} with S = sum_r S_r * exp(M_r - M)."""
        gathered = self.allgather(stats)
        merged = da.empty((2,), stats.dtype)
        _lib.get().lse_merge(gathered._ptr, self.world, merged._ptr, merged._code())
        return merged


class RcclCommunicator(Communicator):
    """RCCL through the C-ABI (tnn_comm_*), collectives enqueued on the library's own stream."""

    def __init__(self, rank, world, unique_id):
        self.rank, self.world = int(rank), int(world)
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _lib.get().comm_init(self.rank, self.world, buf)
        self._open = True
        # RCCL must be torn down before the HIP runtime's own exit handlers run (otherwise the process aborts
        # with "double free or corruption" at interpreter exit)
        atexit.register(self.close)

    @staticmethod
    def new_unique_id():
        buf = ctypes.create_string_buffer(128)
        _lib.get().comm_unique_id(buf)
        return buf.raw

    def allreduce(self, arr, op="sum"):
        code = {"sum": _lib.RSUM, "max": _lib.RMAX, "min": _lib.RMIN}[op]
        _lib.get().allreduce(arr._ptr, arr.size, arr._code(), code)
        return arr

    def allgather(self, arr):
        arr = arr._contig()
        out = da.empty((self.world,) + arr.shape, arr.dtype)
        _lib.get().allgather(arr._ptr, out._ptr, arr.size, arr._code())
        return out

    def barrier(self):
        _lib.synchronize()
        dist = sys.modules.get("torch.distributed")
        if dist is not None and dist.is_available() and dist.is_initialized():
            dist.barrier()

    def close(self):
        if getattr(self, "_open", False):
            self._open = False
            _lib.synchronize()
            _lib.get().comm_destroy()


class GlooCommunicator(Communicator):
    """CPU-test communicator: the same collectives over torch.distributed/gloo via host memory."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allreduce(self, arr, op="sum"):
        import torch
        host = torch.from_numpy(np.ascontiguousarray(np.asarray(arr)))
        red = {"sum": self._dist.ReduceOp.SUM, "max": self._dist.ReduceOp.MAX,
               "min": self._dist.ReduceOp.MIN}[op]
        self._dist.all_reduce(host, op=red)
        arr[...] = da.asarray(host.numpy(), dtype=arr.dtype)
        return arr

    def allgather(self, arr):
        import torch
        host = torch.from_numpy(np.ascontiguousarray(np.asarray(arr)))
        parts = [torch.empty_like(host) for _ in range(self.world)]
        self._dist.all_gather(parts, host)
        return da.asarray(np.stack([p.numpy() for p in parts]), dtype=arr.dtype)

    def barrier(self):
        self._dist.barrier()


def init_from_env(backend="rccl"):
    """Build the communicator for this rank from the torchrun environment (None when world == 1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        if backend == "rccl" and os.environ.get("TNN_FORCE_COMM") == "1":
            # single-GPU boxes: a 1-rank RCCL communicator, so the collectives' code path can be run and timed
            return RcclCommunicator(0, 1, RcclCommunicator.new_unique_id())
        return None
    if backend == "rccl" and _lib.is_loaded() and "torch" not in sys.modules:
        # torch preloads its bundled libamdhip64 / librccl by absolute path: importing it AFTER libtnn_hip.so
        # puts two HIP runtimes in the process (abort at exit, torch.cuda blind).  Imported first, they are the
        # single runtime everything binds to.
        raise RuntimeError("import torch before the first tinynn_autograd_amd device call when running "
                           "data-parallel (see DESIGN.md §7)")
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")          # control plane only
    rank = dist.get_rank()
    if backend == "gloo":
        return GlooCommunicator()
    box = [RcclCommunicator.new_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return RcclCommunicator(rank, world, box[0])
