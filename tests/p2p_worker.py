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
    # which form of the statistics exchange this run takes: deferred into the head launch (every workgroup of which waits for the
    # peers, so all ranks sharing this GPU must fit it together: tnn_mlp_head_bwd_xchg_fits) or at the tail of the forward launch
    pw = trainer._pwidths
    deferred = _xchg_fits(rows, pw[-3], pw[-2], pw[-1])
    want = os.environ.get("TNN_P2P_TEST_EXPECT_XCHG")
    if want is not None:
        assert deferred == (want == "1"), "expected the %s form of the statistics exchange (ranks %d, rows %d)" % (
            "deferred" if want == "1" else "forward-tail", world, rows)
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
    _row_blocked_deferred_exchange(tn, comm, rank, world, dist)
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


def _xchg_fits(rows, n_in, n_hidden, n_classes):
    import ctypes
    from tinynn_autograd_amd import _lib
    fits = ctypes.c_int(0)
    _lib.get().mlp_head_bwd_xchg_fits(rows, n_in, n_hidden, n_classes, _lib.F32, ctypes.byref(fits))
    return bool(fits.value)


def _row_blocked_deferred_exchange(tn, comm, rank, world, dist):
    """The deferred exchange in the ROW-BLOCKED head launch (more than 128 rows per rank: the row-panel forward leaves one pair per
    16-row panel, the head launch merges them, pushes the shard's pair and merges the ranks' pairs) with `world` ranks on one GPU.
    A 16-wide layer in front of the 128 -> 10 head keeps the launch small enough (8 + 16 + rows / 16 workgroups per rank) that the
    ranks fit the GPU together whatever their number — the co-residency rule of tnn_mlp_head_bwd_xchg_fits holds.  Checked against
    the float64 closed-form oracle on the GLOBAL batch: losses to 1e-5, replicas identical."""
    from oracle.closed_form import ClosedFormMLP             # the checker
    from tinynn_autograd_amd.fused import MLPTrainer
    widths, rows, lr, steps = [40, 16, 128, 10], 320, 1e-3, 4
    assert _xchg_fits(rows, 16, 128, 10), "the small row-blocked launch must fit the GPU %d times" % world
    m = rows * world
    rs = np.random.RandomState(77)
    layers = [{"w": rs.uniform(-0.3, 0.3, (a, b)).astype(np.float32), "b": rs.uniform(-0.1, 0.1, (1, b)).astype(np.float32)}
              for a, b in zip(widths[:-1], widths[1:])]
    data = [(rs.uniform(-1, 1, (m, widths[0])).astype(np.float32), np.eye(10, dtype=np.float32)[rs.randint(0, 10, m)])
            for _ in range(steps)]
    oracle = ClosedFormMLP([l["w"] for l in layers], [l["b"] for l in layers], lr=lr)
    dp = MLPTrainer(widths, rows, loss="softmax_nll", optimizer="adam", lr=lr, comm=comm)
    dp.set_parameters(layers)
    sl = slice(rank * rows, (rank + 1) * rows)
    for s, (x, y) in enumerate(data):
        got = float(dp.step(tn.asarray(x[sl]), tn.asarray(y[sl])))
        want = oracle.step(x, y)[0]
        np.testing.assert_allclose(got, want, rtol=1e-5, err_msg="row-blocked deferred exchange, step %d" % s)
    flat = np.asarray(dp.params)
    parts = [None] * world
    dist.all_gather_object(parts, flat.tobytes())
    assert all(p == parts[0] for p in parts), "parameters diverged across ranks (row-blocked deferred exchange)"
    assert not comm.p2p_status()["dead"]


if __name__ == "__main__":
    main()
