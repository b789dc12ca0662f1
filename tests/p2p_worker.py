"""Worker of tests/test_gpu_p2p.py: one rank of a world_size-N data-parallel run whose collectives go through the
xGMI peer-to-peer transport (csrc/tnn_p2p.hip).  The GPU box has ONE GPU, so all ranks share device 0: IPC mapping,
the flag protocol, buffer reuse across calls, hipGraph replay and the timeout path are exercised for real; what a
single GPU cannot show is the cross-DEVICE visibility of the uncached regions (covered by design, see the kernel
header, and by the start-up self-test every multi-GPU run performs before trusting the path).

Checks: self-test (bit-exact sums in rank order), sharded bs=1024 trajectory against the reference fixture
(tests/golden/traj_D_adam.npz) eagerly and replayed from one hipGraph, identical parameters on every rank."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch                        # noqa: F401  (first: one HIP runtime per process)
    import torch.distributed as dist
    import tinynn_autograd_amd as tn
    import helpers as H
    from tinynn_autograd_amd.dist import XgmiCommunicator
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = XgmiCommunicator(rank, world, p2p_bytes=4 << 20)
    assert comm.p2p_status() == {"connected": True, "enabled": True, "dead": False}
    assert comm.p2p_selftest(sizes=(1, 5, 1000, 235147, 1 << 20), rounds=3), "self-test failed"

    # TNN_P2P_TEST_TRAJ: "D_adam" (bs 1024: 512 rows per rank at world 2 — the 8-launch step with the exchange in the
    # one-workgroup loss kernel) or "A_adam" (bs 128: 64 rows per rank — the 5-launch step whose multi-workgroup head
    # exchanges the statistics inside the launch)
    cfg, gold = H.load_traj(os.environ.get("TNN_P2P_TEST_TRAJ", "D_adam"))
    w, m = cfg["widths"], cfg["m"]
    rows = m // world
    sl = slice(rank * rows, (rank + 1) * rows)
    model, _ = H.build_model(cfg)
    trainer = tn.trainer_from_net(model.net, max_rows=rows, loss="softmax_nll", optimizer="adam", lr=cfg["lr"], comm=comm)
    data = list(H.batches(cfg["data_seed"], 5, m, w[0], w[-1], cfg["loss"]))
    for s in range(3):                                                      # eager sharded steps
        x, y = data[s]
        tl = float(trainer.step(tn.asarray(x[sl]), tn.asarray(y[sl])))
        np.testing.assert_allclose(tl, gold["loss"][s], rtol=1e-5, err_msg="eager step %d" % s)
    graph = trainer.capture_steps([(tn.asarray(x[sl]), tn.asarray(y[sl])) for x, y in data[3:5]])
    losses = np.asarray(graph.launch())                                     # two more steps, one hipGraphLaunch
    np.testing.assert_allclose(losses, gold["loss"][3:5], rtol=1e-5, err_msg="captured steps")
    flat = np.asarray(trainer.params)
    parts = [None] * world
    dist.all_gather_object(parts, flat.tobytes())
    assert all(p == parts[0] for p in parts), "parameters diverged across ranks"
    assert not comm.p2p_status()["dead"]
    # many back-to-back calls: buffer reuse without host synchronisation in between
    buf = tn.asarray(np.full(50000, 1.0 + rank, np.float32))
    for _ in range(200):
        comm.allreduce(buf)
        buf *= 1.0 / world
    want = np.float32(sum(1.0 + r for r in range(world))) / np.float32(world)
    np.testing.assert_allclose(np.asarray(buf), want, rtol=1e-5)
    comm.barrier()
    # a peer that never shows up must not hang the GPU and must not corrupt training: rank 0 alone enters one more
    # sharded training step (loss exchange + all-reduce with the Adam tail).  Its barriers time out (TNN_P2P_TIMEOUT_MS),
    # the kernels finish, the sticky `dead` word and its host mirror report it — and parameters, Adam moments and the
    # beta powers' consumers are left exactly as they were (partial sums are never applied).
    if os.environ.get("TNN_P2P_TEST_TIMEOUT") == "1":
        from tinynn_autograd_amd import _lib
        from tinynn_autograd_amd.dist import PeerTimeout
        if rank == 0:
            import time
            before = [np.asarray(a).copy() for a in (trainer.params, trainer.adam_m, trainer.adam_v)]
            x, y = data[0]
            t0 = time.time()
            trainer.step(tn.asarray(x[sl]), tn.asarray(y[sl]))
            st = comm.p2p_status()                   # synchronises the stream
            assert st["dead"] and comm.p2p_failed(), "timeout was not reported"
            assert time.time() - t0 < 10.0
            after = [np.asarray(a) for a in (trainer.params, trainer.adam_m, trainer.adam_v)]
            assert all(np.array_equal(a, b) for a, b in zip(before, after)), "a timed-out step changed parameters / Adam state"
            for call in (lambda: comm.allreduce(buf), lambda: trainer.step(tn.asarray(x[sl]), tn.asarray(y[sl])),
                         lambda: graph.launch()):
                try:                                 # later calls into the dead transport fail loudly
                    call()
                    raise AssertionError("a call into the dead transport must raise")
                except (_lib.TnnError, PeerTimeout):
                    pass
            after = [np.asarray(a) for a in (trainer.params, trainer.adam_m, trainer.adam_v)]
            assert all(np.array_equal(a, b) for a, b in zip(before, after))
        # barrier() mirrors the dead word collectively: EVERY rank learns of it, the transport goes off everywhere
        try:
            comm.barrier()
            raise AssertionError("barrier() must raise PeerTimeout on every rank after a timeout on any rank")
        except PeerTimeout:
            pass
        assert not comm.p2p_status()["enabled"]
    comm.close()
    print("p2p_worker rank %d/%d ok" % (rank, world))


if __name__ == "__main__":
    main()
