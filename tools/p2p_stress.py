"""Stress the xGMI peer-to-peer collectives (csrc/tnn_p2p.hip) with W ranks sharing GPU 0: thousands of all-reduces
of random sizes whose inputs every rank can reproduce, checked BIT-EXACTLY every iteration, interleaved with small
all-gathers and, every few iterations, un-synchronised bursts (no host sync between calls) so buffer reuse and the
epoch logic are exercised back to back.  Every third iteration a data-parallel TRAINING step runs in between (the launch
that carries the first layer's backward, the gradient all-reduce and Adam: it shares the slots and the per-workgroup tags with
the plain all-reduce), on different rows per rank; afterwards the replicas' parameters must be bit-identical (a checksum of
their bit patterns, all-gathered).
  python tools/p2p_stress.py --spawn 4 --iters 3000
Debugging switches: STRESS_SIZE=<floats> (one message size only), STRESS_NO_BURST=1 (a host sync after every all-reduce),
STRESS_NO_STEPS=1 (no training steps in between); on failure every rank prints the transport's debug words."""
import argparse
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def contribution(np, r, n, it):
    i = np.arange(n, dtype=np.int64)
    return (((i * 7 + r * 13 + n + 31 * it) % 1009).astype(np.float32) / np.float32(1009.0)) - np.float32(0.37 * (r % 3))


def worker(args):
    import numpy as np
    import torch  # noqa: F401
    import torch.distributed as dist
    import tinynn_autograd_amd as tn
    from tinynn_autograd_amd.dist import XgmiCommunicator
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    rank = dist.get_rank() if world > 1 else 0
    comm = XgmiCommunicator(rank, world, p2p_bytes=4 << 20)
    rs = np.random.RandomState(99)                       # same stream on every rank
    bad = 0
    from tinynn_autograd_amd.fused import MLPTrainer
    widths, rows = [784, 256, 128, 10], 64
    init = [{"w": rs.uniform(-0.1, 0.1, (a, b)).astype(np.float32), "b": rs.uniform(-0.1, 0.1, (1, b)).astype(np.float32)}
            for a, b in zip(widths[:-1], widths[1:])]
    trainer = MLPTrainer(widths, rows, loss="softmax_nll", optimizer="adam", lr=1e-3, comm=comm, force_dp=True)
    trainer.set_parameters(init)
    rows_rng = np.random.RandomState(1000 + rank)        # this rank's rows
    steps = 0
    for it in range(args.iters):
        if it % 3 == 0 and not os.environ.get("STRESS_NO_STEPS"):
            x = tn.asarray(rows_rng.uniform(-1, 1, (rows, widths[0])).astype(np.float32))
            y = tn.asarray(np.eye(10, dtype=np.float32)[rows_rng.randint(0, 10, rows)])
            trainer.step(x, y)
            steps += 1
            if it % 30 == 0:
                p = np.asarray(trainer.flat_parameters())
                h = int(np.ascontiguousarray(p).view(np.uint32).astype(np.uint64).sum())       # checksum of the bit patterns
                words = tn.asarray(np.array([(h >> (16 * k)) & 0xffff for k in range(4)], np.float32))
                got = np.asarray(comm.allgather(words))
                if not (got == got[0]).all() or not np.isfinite(p).all():
                    bad += 1
        n = int(rs.choice([1, 3, 17, 257, 1000, 4099, 29400, 65536, 235147, 500001, 1 << 20]))
        if os.environ.get("STRESS_SIZE"):                # one size only (debugging)
            n = int(os.environ["STRESS_SIZE"])
        burst = 1 + int(rs.randint(0, 4)) if it % 5 == 0 else 1
        if os.environ.get("STRESS_NO_BURST"):
            burst = 1
        bufs, wants = [], []
        for b in range(burst):
            parts = [contribution(np, r, n, it * 8 + b) for r in range(world)]
            want = parts[0].copy()
            for r in range(1, world):
                want = want + parts[r]
            bufs.append(tn.asarray(parts[rank]))
            wants.append(want)
        for buf in bufs:                                 # enqueue the whole burst before looking at anything
            comm.allreduce(buf)
        for buf, want in zip(bufs, wants):
            if not np.array_equal(np.asarray(buf), want):
                bad += 1
        if it % 7 == 0:
            mine = tn.asarray(np.array([rank + it, -it, 0.5 * rank, 3.0], np.float32))
            got = np.asarray(comm.allgather(mine))
            want = np.array([[r + it, -it, 0.5 * r, 3.0] for r in range(world)], np.float32)
            if not np.array_equal(got, want):
                bad += 1
    st = comm.p2p_status()
    comm.barrier()
    comm.close()
    print("rank %d/%d: %d iterations (%d training steps in between), %d mismatches, dead=%s" % (rank, world, args.iters, steps, bad,
                                                                                             st["dead"]), flush=True)
    sys.exit(1 if bad or st["dead"] else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spawn", type=int, default=0)
    ap.add_argument("--iters", type=int, default=2000)
    args = ap.parse_args()
    if args.spawn > 1:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(args.spawn):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.spawn), TNN_DEVICE="0", MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--iters", str(args.iters)], env=env))
        sys.exit(max(p.wait() for p in procs))
    try:
        worker(args)
    except Exception:
        import ctypes
        from tinynn_autograd_amd import _lib
        w = (ctypes.c_int * 16)()
        _lib.get().p2p_debug(w)
        print("rank %s: failed; the transport's debug words (which wait gave up, expected, seen, peer / workgroup, detail): %s"
              % (os.environ.get("RANK", "0"), [int(x) & 0xffffffff for x in list(w)[:5]]), flush=True)
        raise


if __name__ == "__main__":
    main()
