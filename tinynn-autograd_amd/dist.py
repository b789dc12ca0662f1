"""Data-parallel plumbing: one process per GPU; data plane = RCCL plus an xGMI peer-to-peer latency path.

New relative to the reference (no communication there).  Two collectives exist on the path:
  C1  in-place SUM all-reduce of the flat gradient arena between backward() and the optimizer step;
  C2  an all-gather of one {max, sum-exp} pair per rank inside the whole-batch softmax loss
      (core/losses.py:26-27 couples the shards), merged with a log-sum-exp kernel.

Both go through tnn_allreduce / tnn_allgather: small f32 sums and tiny all-gathers take the peer-to-peer transport
(csrc/tnn_p2p.hip: IPC-mapped uncached regions, pushed stores, flag barriers — self-tested at start-up, see
`_try_p2p`), everything else RCCL.

`torch.distributed` (gloo) is used only as the control plane: rendezvous from the RANK / WORLD_SIZE /
MASTER_* environment that `python -m torch.distributed.run` provides, broadcast of the RCCL unique id, exchange
of the IPC handles, barriers and the max-over-ranks of bench timings.  Import torch BEFORE the first device call of this package
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


class PeerTimeout(RuntimeError):
    """A peer never reached an xGMI peer-to-peer barrier (TNN_P2P_TIMEOUT_MS).  The collective that timed out and every
    one issued since were DISCARDED on the device: gradient buffers, parameters and optimizer state still hold what
    they held before it, on every rank that saw the timeout.  The transport has been switched off on all ranks (RCCL
    carries the collectives from here on, when a communicator exists); redo the step."""


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


def _control_plane():
    dist = sys.modules.get("torch.distributed")
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist
    return None


class DeviceCommunicator(Communicator):
    """Collectives through the C-ABI (tnn_allreduce / tnn_allgather) on the library's own stream.  Two transports
    sit under those entry points: RCCL (any size / dtype) and the xGMI peer-to-peer path of csrc/tnn_p2p.hip (f32
    sums up to `p2p_bytes`, all-gathers up to 256 B per rank) which the library prefers whenever it is mapped."""

    def __init__(self, rank, world):
        self.rank, self.world = int(rank), int(world)
        self._rccl = False
        self._p2p = False
        self.p2p_bytes = 0                # capacity of the mapped peer regions (largest f32 sum they carry)
        self._open = True
        # tear the transports down before the HIP runtime's own exit handlers run (RCCL otherwise aborts the
        # process with "double free or corruption" at interpreter exit)
        atexit.register(self.close)

    # ---- transports
    def enable_rccl(self, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _lib.get().comm_init(self.rank, self.world, buf)
        self._rccl = True
        return self

    def enable_p2p(self, max_bytes=8 << 20, exchange=None, bulk_bytes=0):
        """Create this rank's region, swap the 64-byte IPC handles (`exchange(bytes) -> [bytes] * world`, default:
        torch.distributed.all_gather_object on the gloo control plane) and map every peer.  bulk_bytes > 0 also reserves staging
        for bandwidth-sized reduce-scatter / all-gather over the same regions (tnn_p2p_set_bulk_bytes: what a group WITHOUT an
        RCCL communicator needs for the bf16 trainer's sharded-optimizer step)."""
        lib = _lib.get()
        mine = ctypes.create_string_buffer(64)
        created, failure = False, None
        try:
            lib.p2p_set_bulk_bytes((int(bulk_bytes) + 4095) // 4096 * 4096)
            lib.p2p_create(self.rank, self.world, int(max_bytes), mine)
            created = True
        except Exception as e:                  # noqa: BLE001 - still take part in the exchange below
            failure = e
        token = mine.raw if created else None
        # every rank takes part in the exchange even if its own create failed, so nobody waits forever
        if exchange is not None:
            handles = exchange(token)
        elif self.world == 1:
            handles = [token]
        else:
            dist = _control_plane()
            if dist is None:
                if created:
                    lib.p2p_destroy()
                raise RuntimeError("enable_p2p: torch.distributed is not initialised and no exchange() was given")
            handles = [None] * self.world
            dist.all_gather_object(handles, token)
        try:
            if failure is not None:
                raise failure
            if any(h is None for h in handles):
                raise RuntimeError("enable_p2p: a peer could not create its region")
            blob = ctypes.create_string_buffer(b"".join(bytes(h) for h in handles), 64 * self.world)
            lib.p2p_connect(blob)
        except Exception:
            if created:
                lib.p2p_destroy()
            raise
        self._p2p = True
        self.p2p_bytes = int(max_bytes)
        self.p2p_bulk_bytes = int(bulk_bytes)
        return self

    def set_p2p(self, on):
        if self._p2p:
            _lib.get().p2p_enable(1 if on else 0)

    def p2p_status(self):
        """{'connected', 'enabled', 'dead'}; dead = a peer barrier timed out (synchronises the stream)."""
        c, e, d = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.get().p2p_status(ctypes.byref(c), ctypes.byref(e), ctypes.byref(d))
        st = {"connected": bool(c.value), "enabled": bool(e.value), "dead": bool(d.value)}
        if d.value:                  # which wait gave up: 1 collective flag barrier, 2 statistics exchange, 3 hand-over row
            w = (ctypes.c_int * 16)()
            _lib.get().p2p_debug(w)
            st["dead_code"] = int(d.value)
            st["first_timeout"] = {"wait": w[0], "expected": w[1] & 0xffffffff, "seen": w[2] & 0xffffffff, "peer_or_workgroup": w[3],
                                   "detail": w[4]}
        return st

    def p2p_failed(self):
        """True once a peer barrier has timed out — read from the host-pinned mirror of the device's sticky word, no
        stream synchronisation (the native calls into the transport return errors from then on anyway; this lets a
        graph-replay loop look before it launches)."""
        if not self._p2p:
            return False
        f = ctypes.c_int(0)
        _lib.get().p2p_poll_failed(ctypes.byref(f))
        return bool(f.value)

    def check(self, collective=True):
        """Mirror the transport's `dead` word to the host at a synchronisation point.  collective=True (every rank
        calls it — barrier() does): the ranks vote, and if ANY of them saw a timeout the transport is switched off on
        ALL of them and PeerTimeout is raised everywhere, so replicas cannot go on with different ideas of which
        transport is live.  collective=False: local look only."""
        failed = self.p2p_failed()
        if collective and self.world > 1:
            dist = _control_plane()
            if dist is not None:
                import torch
                t = torch.tensor([1 if failed else 0])
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                failed = bool(int(t.item()))
        if failed:
            sys.stderr.write("[tinynn_autograd_amd] rank %d transport status at the failed check: %s\n" % (self.rank, self.p2p_status()))
            self.set_p2p(False)
            raise PeerTimeout("xGMI peer-to-peer barrier timed out on rank %d or a peer; collectives since then were "
                              "discarded (parameters untouched); the transport is now off on every rank%s"
                              % (self.rank, " and RCCL takes over" if self._rccl else " and no RCCL communicator exists"))

    def p2p_selftest(self, sizes=(1, 1000, 235147, 65536), rounds=3):
        """All-reduce inputs every rank can reproduce locally (x_r[i] is a function of r and i), so the expected
        sum — float32 adds in rank order, exactly what the kernel does — needs no second transport: the comparison is
        bit-exact.  Also runs the small all-gather.  Returns True when every round matched and no barrier timed out."""
        ok = True
        for n in sizes:
            i = np.arange(n, dtype=np.int64)
            for k in range(rounds):                 # fresh values every round: a stale read cannot pass
                contrib = [(((i * 7 + r * 13 + n + 31 * k) % 101).astype(np.float32) / np.float32(101.0))
                           - np.float32(0.5 * (r % 2)) for r in range(self.world)]
                want = contrib[0].copy()
                for r in range(1, self.world):
                    want = want + contrib[r]
                buf = da.asarray(contrib[self.rank])
                self.allreduce(buf)
                ok = ok and np.array_equal(np.asarray(buf), want)
        for k in range(rounds):
            mine = da.asarray(np.array([self.rank + 0.25 * k, -1.0 - self.rank], np.float32))
            got = np.asarray(self.allgather(mine))
            want = np.array([[r + 0.25 * k, -1.0 - r] for r in range(self.world)], np.float32)
            ok = ok and np.array_equal(got, want)
        # the deferred statistics exchange of the data-parallel head launch (every workgroup of a 64-workgroup launch merges the
        # ranks' pairs: tnn_p2p_xchg_selftest) against the same merge on the host
        for k in range(rounds):
            pairs = [(np.float32(0.25 * r + 0.5 * k), np.float32(1.0 + 0.5 * r + k)) for r in range(self.world)]
            out = da.empty((64, 2), np.float32)
            _lib.get().p2p_xchg_selftest(float(pairs[self.rank][0]), float(pairs[self.rank][1]), out._ptr)
            big = max(float(m) for m, _ in pairs)
            want = (big, sum(float(s_) * float(np.exp(np.float32(float(m) - big))) for m, s_ in pairs))
            got = np.asarray(out, dtype=np.float64)
            ok = ok and bool(np.all(got[:, 0] == want[0]) and np.allclose(got[:, 1], want[1], rtol=1e-5, atol=0.0))
        # ... and, on a group without an RCCL communicator, the bulk path (reduce-scatter with fp32 accumulation in rank order, all-gather)
        if getattr(self, "p2p_bulk_bytes", 0) > 0 and not self._rccl and self.world > 1:
            n = 4096 + 8
            contrib = [np.random.RandomState(7 + q).uniform(-1, 1, n * self.world).astype(np.float32) for q in range(self.world)]
            send, recv = da.asarray(contrib[self.rank]), da.zeros((n,), np.float32)
            _lib.get().reduce_scatter(send._ptr, recv._ptr, n, _lib.F32)
            ok = ok and np.array_equal(np.asarray(recv), _rank_order_sum(contrib, self.rank, n))
            whole = da.zeros((n * self.world,), np.float32)
            _lib.get().allgather(recv._ptr, whole._ptr, n, _lib.F32)
            want = np.concatenate([_rank_order_sum(contrib, r, n) for r in range(self.world)])
            ok = ok and np.array_equal(np.asarray(whole), want)
        return bool(ok and not self.p2p_status()["dead"])

    # ---- collectives
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
        dist = _control_plane()
        if dist is not None:
            dist.barrier()
        if self._p2p and self._open:
            self.check(collective=True)

    def close(self):
        if getattr(self, "_open", False):
            self._open = False
            _lib.synchronize()
            if self._p2p:
                # nobody may unmap while a peer can still store into the region
                try:
                    dist = _control_plane()
                    if dist is not None and self.world > 1:
                        dist.barrier()
                except Exception:
                    pass
                _lib.get().p2p_destroy()
                self._p2p = False
            if self._rccl:
                _lib.get().comm_destroy()
                self._rccl = False


def _rank_order_sum(contrib, r, n):
    acc = contrib[0][r * n:(r + 1) * n].copy()
    for q in range(1, len(contrib)):
        acc = acc + contrib[q][r * n:(r + 1) * n]
    return acc


class RcclCommunicator(DeviceCommunicator):
    """RCCL communicator (tnn_comm_*); `p2p=True` additionally maps the peers for the xGMI latency path."""

    def __init__(self, rank, world, unique_id, p2p=False, p2p_bytes=8 << 20):
        DeviceCommunicator.__init__(self, rank, world)
        self.enable_rccl(unique_id)
        if p2p:
            self.enable_p2p(p2p_bytes)

    @staticmethod
    def new_unique_id():
        buf = ctypes.create_string_buffer(128)
        _lib.get().comm_unique_id(buf)
        return buf.raw


class XgmiCommunicator(DeviceCommunicator):
    """Peer-to-peer transport only (no RCCL communicator): f32 sums up to `p2p_bytes` and small all-gathers."""

    def __init__(self, rank, world, p2p_bytes=8 << 20, exchange=None, bulk_bytes=0):
        DeviceCommunicator.__init__(self, rank, world)
        self.enable_p2p(p2p_bytes, exchange, bulk_bytes)


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


def _want_p2p(p2p):
    if p2p is None:
        return os.environ.get("TNN_P2P", "1") != "0"
    return bool(p2p)


def _try_p2p(comm, dist, bulk_bytes=0):
    """Map the peers and prove the path before trusting it: any failure — IPC refused, a wrong sum, a barrier
    timeout — on ANY rank leaves every rank on RCCL.  Returns True when the peer-to-peer path is live."""
    import torch
    ok = 1
    try:
        comm.enable_p2p(int(os.environ.get("TNN_P2P_BYTES", str(8 << 20))), bulk_bytes=bulk_bytes)
    except Exception as e:                      # noqa: BLE001 - every failure means "stay on RCCL"
        sys.stderr.write("[tinynn_autograd_amd] xGMI peer-to-peer path unavailable on rank %d: %s\n" % (comm.rank, e))
        ok = 0
    flag = torch.tensor([ok])
    if dist is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag) == 0:
        if comm._p2p:
            comm.set_p2p(False)
        return False
    try:
        good = 1 if comm.p2p_selftest() else 0
    except Exception as e:                      # noqa: BLE001 - keep every rank in the vote below
        sys.stderr.write("[tinynn_autograd_amd] xGMI peer-to-peer self-test raised on rank %d: %s\n" % (comm.rank, e))
        good = 0
    flag = torch.tensor([good])
    if dist is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag) == 0:
        sys.stderr.write("[tinynn_autograd_amd] xGMI peer-to-peer self-test failed on rank %d; using RCCL\n" % comm.rank)
        comm.set_p2p(False)
        return False
    return True


def init_from_env(backend="rccl", p2p=None):
    """Build the communicator for this rank from the torchrun environment (None when world == 1).
    backend "rccl": RCCL plus, unless p2p=False / TNN_P2P=0, the self-tested xGMI peer-to-peer latency path."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        if backend == "rccl" and os.environ.get("TNN_FORCE_COMM") == "1":
            # single-GPU boxes: a 1-rank RCCL communicator, so the collectives' code path can be run and timed
            comm = RcclCommunicator(0, 1, RcclCommunicator.new_unique_id())
            if _want_p2p(p2p):
                import torch  # noqa: F401
                _try_p2p(comm, None)
            return comm
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
    if os.environ.get("TNN_COMM") == "xgmi":
        # peer-to-peer transport only, no RCCL communicator: RCCL refuses ranks that share a GPU, this does not —
        # how the N > 1 code paths (bench.py included) are exercised on a one-GPU box (TNN_DEVICE=0 for every rank)
        comm = DeviceCommunicator(rank, world)
        # no RCCL here: bandwidth-sized reduce-scatter / all-gather (the bf16 trainer's sharded optimizer) go over the mapped
        # regions too — 4 MiB of staging per (parity, source) unless TNN_P2P_BULK_BYTES says otherwise
        if not _try_p2p(comm, dist, bulk_bytes=int(os.environ.get("TNN_P2P_BULK_BYTES", str(4 << 20)))):
            raise RuntimeError("TNN_COMM=xgmi: the peer-to-peer transport is not available")
        return comm
    box = [RcclCommunicator.new_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = RcclCommunicator(rank, world, box[0])
    if _want_p2p(p2p):
        _try_p2p(comm, dist)
    return comm
