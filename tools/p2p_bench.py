"""Latency of the xGMI peer-to-peer collectives (csrc/tnn_p2p.hip), replayed from a hipGraph.
  python tools/p2p_bench.py                      one rank
  python tools/p2p_bench.py --spawn 2            two processes sharing GPU 0 (what a 1-GPU box allows)
Under torchrun on a multi-GPU node it measures the real thing (one rank per GPU)."""
import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import numpy as np
    import torch  # noqa: F401
    import torch.distributed as dist
    import tinynn_autograd_amd as tn
    from tinynn_autograd_amd import _lib
    from tinynn_autograd_amd.dist import XgmiCommunicator
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    rank = dist.get_rank() if world > 1 else 0
    comm = XgmiCommunicator(rank, world, p2p_bytes=8 << 20)
    assert comm.p2p_selftest(sizes=(1000, 235147), rounds=1)
    for n in [int(v) for v in args.sizes.split(",")]:
        buf = tn.asarray(np.zeros(n, np.float32))
        reps = 200
        g = _lib.Graph()
        with g:
            for _ in range(reps):
                comm.allreduce(buf)
        g.launch(); comm.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            g.launch()
        _lib.synchronize()
        dt = (time.perf_counter() - t0) / (5 * reps)
        if rank == 0:
            print("allreduce n=%8d floats  %7.2f us" % (n, dt * 1e6), flush=True)
        comm.barrier()
    st = tn.asarray(np.array([1.0, 2.0], np.float32))
    out = tn.empty((world, 2))
    g = _lib.Graph()
    with g:
        for _ in range(200):
            _lib.get().allgather(st._ptr, out._ptr, 2, _lib.F32)
    g.launch(); comm.barrier()
    t0 = time.perf_counter()
    for _ in range(5):
        g.launch()
    _lib.synchronize()
    if rank == 0:
        print("allgather 2 floats/rank      %7.2f us" % ((time.perf_counter() - t0) / 1000 * 1e6), flush=True)
    assert not comm.p2p_status()["dead"]
    comm.barrier()
    comm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spawn", type=int, default=0)
    ap.add_argument("--sizes", default="1024,29400,235147,1048576")
    args = ap.parse_args()
    if args.spawn > 1:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(args.spawn):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.spawn), TNN_DEVICE="0", MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sizes", args.sizes], env=env))
        sys.exit(max(p.wait() for p in procs))
    worker(args)


if __name__ == "__main__":
    main()
