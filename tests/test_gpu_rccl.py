"""RCCL through the C-ABI on the GPU box.  The box has ONE GPU, so the communicator has one rank: this checks
library loading (dlopen of ROCm's librccl), unique-id / init / collectives / destroy, and that the sharded
trainer and Model code paths (stats all-gather + merge, arena all-reduce) reproduce the single-process
fixtures when every collective is the identity.  World sizes > 1 are covered on CPU by test_dist_gloo.py."""

import numpy as np
import pytest

import helpers as H
import tinynn_autograd_amd as tn
from tinynn_autograd_amd.core.tensor import Tensor


@pytest.fixture(scope="module")
def comm():
    from tinynn_autograd_amd.dist import RcclCommunicator
    c = RcclCommunicator(0, 1, RcclCommunicator.new_unique_id())
    yield c
    c.close()


@pytest.mark.gpu
def test_rccl_collectives_world1(comm):
    x = np.arange(1000, dtype=np.float32)
    d = tn.asarray(x)
    comm.allreduce(d)
    assert np.array_equal(np.asarray(d), x)
    comm.allreduce(d, op="max")
    assert np.array_equal(np.asarray(d), x)
    g = comm.allgather(tn.asarray(np.array([1.5, 2.5], dtype=np.float32)))
    assert g.shape == (1, 2) and np.array_equal(np.asarray(g), [[1.5, 2.5]])
    merged = comm.merge_softmax_stats(tn.asarray(np.array([0.25, 7.0], dtype=np.float32)))
    np.testing.assert_allclose(np.asarray(merged), [0.25, 7.0], rtol=1e-6)


@pytest.mark.gpu
def test_sharded_paths_with_rccl_world1(comm):
    cfg, gold = H.load_traj("A_adam")
    w = cfg["widths"]
    model, loss_layer = H.build_model(cfg, comm=comm)
    loss_layer.comm = comm
    trainer = tn.trainer_from_net(model.net, max_rows=cfg["m"], lr=cfg["lr"], comm=comm, force_dp=True)
    assert trainer.comm is comm
    for s, (x, y) in enumerate(H.batches(cfg["data_seed"], 6, cfg["m"], w[0], w[-1], cfg["loss"])):
        model.zero_grad()
        loss = loss_layer.loss(model.forward(Tensor(x)), Tensor(y))
        loss.backward()
        if model._grad_arena is not None:
            comm.allreduce(model._grad_arena)               # what Model.step does when world > 1
        model.step()
        np.testing.assert_allclose(float(loss.values), gold["loss"][s], rtol=1e-5)
        tl = float(trainer.step(tn.asarray(x), tn.asarray(y)))
        np.testing.assert_allclose(tl, gold["loss"][s], rtol=1e-5)


@pytest.mark.gpu
def test_sharded_steps_captured_with_rccl_in_graph(comm):
    """The data-parallel step including its two RCCL collectives replayed from ONE hipGraph."""
    cfg, gold = H.load_traj("A_adam")
    w = cfg["widths"]
    model, _ = H.build_model(cfg)
    trainer = tn.trainer_from_net(model.net, max_rows=cfg["m"], lr=cfg["lr"], comm=comm, force_dp=True)
    data = H.batches(cfg["data_seed"], cfg["steps"], cfg["m"], w[0], w[-1], cfg["loss"])
    graph = trainer.capture_steps([(tn.asarray(x), tn.asarray(y)) for x, y in data])
    losses = np.asarray(graph.launch())
    np.testing.assert_allclose(losses, gold["loss"], rtol=1e-5)
    for l in range(trainer.n_layers):
        H.check_summary(np.asarray(trainer.param_view(l, "w")), gold, "final_%dw" % l, rtol=0, atol=0.01 * cfg["lr"])       # (parity_suite.ADAM_GATE)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1024, 512, 128])
def test_reference_example_net_sharded_step_world1(comm, rows):
    """The reference's OWN example net (examples/mnist/run.py:59-69) through the data-parallel step at world 1 on RCCL: the
    merged 2L - 2 launch form (forward tail reducing the statistics behind an arrival counter, all-gather of the pair, the
    generic merged head taking the pair from memory in blocks of 128 rows, first-layer backward + all-reduce + Adam) — losses
    against the reference's bs-1024 trajectory (tests/golden/traj_R_example_D.npz) at 1024 rows, against the unsharded
    trainer at 512 / 128 rows per rank (the per-rank batches of 2 / 8 ranks), eager and replayed from a hipGraph."""
    cfg, gold = H.load_traj("R_example_D")
    w = cfg["widths"]
    model, _ = H.build_model(cfg)
    dp = tn.trainer_from_net(model.net, max_rows=rows, lr=cfg["lr"], comm=comm, force_dp=True)
    assert dp.comm is comm and dp.padded and dp._pwidths == [784, 208, 112, 80, 32, 10]
    model2, _ = H.build_model(cfg)
    solo = tn.trainer_from_net(model2.net, max_rows=rows, lr=cfg["lr"])
    data = [(tn.asarray(x[:rows]), tn.asarray(y[:rows])) for x, y in H.batches(cfg["data_seed"], 5, cfg["m"], w[0], w[-1], cfg["loss"])]
    got = [float(dp.step(x, y)) for x, y in data[:3]]
    graph = dp.capture_steps(data[3:])
    got += [float(v) for v in np.asarray(graph.launch())]
    want = [float(solo.step(x, y)) for x, y in data]
    np.testing.assert_allclose(got, want, rtol=1e-5)
    if rows == cfg["m"]:
        np.testing.assert_allclose(got, gold["loss"], rtol=1e-5)
    np.testing.assert_allclose(np.asarray(dp.flat_parameters()), np.asarray(solo.flat_parameters()), rtol=1e-5, atol=5e-5)


@pytest.mark.gpu
def test_bf16_trainer_sharded_step_world1(comm):
    """bf16 trainer through tnn_mlp_step_sharded at world 1 = the sharded-optimizer step (mlp16_step_zero): reduce-scatter of
    the bf16 weight gradient / Adam on the owned rows (all of them here) / all-gather of the bf16 rows, each layer's chain on
    the communication stream.  Against the unsharded step, which keeps the gradient in fp32: the first loss is identical
    (same forward), afterwards the one extra rounding of dW to bf16 shows (Adam's update is sign-like: |step| <= lr)."""
    from tinynn_autograd_amd import bf16
    from tinynn_autograd_amd.fused import MLPTrainer
    rs = np.random.RandomState(5)
    widths, m, lr = [256, 256, 256], 128, 1e-3
    a = np.sqrt(6.0 / 512)
    layers = [{"w": rs.uniform(-a, a, (256, 256)).astype(np.float32), "b": np.zeros((1, 256), np.float32)}
              for _ in range(2)]
    x16 = bf16.to_bf16(rs.rand(m, 256).astype(np.float32))
    losses, params, w16 = [], [], []
    for mode in ("plain", "sharded", "sharded_graph"):
        c = None if mode == "plain" else comm
        t = MLPTrainer(widths, m, loss="mse", optimizer="adam", lr=lr, dtype="bfloat16", comm=c, force_dp=c is not None)
        t.set_parameters(layers)
        if mode == "sharded_graph":
            g = t.capture_steps([(x16, x16)] * 3)
            losses.append([float(v) for v in np.asarray(g.launch())])
        else:
            losses.append([float(t.step(x16, x16)) for _ in range(3)])
        params.append(np.asarray(t.params).copy())
        w16.append(np.asarray(t.weights_bf16()).copy())
        # the bf16 working copy is the rounding of the fp32 master weights
        assert np.array_equal(bf16.round_to_bf16(params[-1]), (w16[-1].astype(np.uint32) << 16).view(np.float32))
        # and the next forward sees it (W^T refreshed behind the all-gather)
        out = np.asarray(t.forward(x16), dtype=np.float64)
        W = [(np.asarray(t.weights_bf16(l)).astype(np.uint32) << 16).view(np.float32).astype(np.float64) for l in range(2)]
        B = [np.asarray(t.param_view(l, "b"), dtype=np.float64) for l in range(2)]
        xs = np.asarray(bf16.to_f32(x16), dtype=np.float64)
        ref = np.clip(xs @ W[0] + B[0], 0, None)
        ref = bf16.round_to_bf16(ref.astype(np.float32)).astype(np.float64) @ W[1] + B[1]
        assert np.abs(out - ref).max() <= 2e-2 * np.abs(ref).max()
    assert losses[0][0] == losses[1][0]
    np.testing.assert_allclose(losses[1], losses[0], rtol=2e-3)
    err = np.abs(params[1] - params[0])
    assert err.max() <= 2 * 3 * lr and np.median(err) <= 0.1 * lr
    # eager and captured sharded steps are the same launches
    assert losses[2] == losses[1] and np.array_equal(params[2], params[1]) and np.array_equal(w16[2], w16[1])


@pytest.mark.gpu
def test_float64_trainer_sharded_step_world1(comm):
    """float64 arenas through the sharded step (RCCL carries f64; the f32-only peer-to-peer path must step aside)."""
    cfg, gold = H.load_traj("A_adam")
    w = cfg["widths"]
    tn.set_default_float(np.float64)
    try:
        model, _ = H.build_model(cfg)
        trainer = tn.trainer_from_net(model.net, max_rows=cfg["m"], lr=cfg["lr"], comm=comm, force_dp=True,
                                      dtype=np.float64)
        for s, (x, y) in enumerate(H.batches(cfg["data_seed"], 5, cfg["m"], w[0], w[-1], cfg["loss"])):
            tl = float(trainer.step(tn.asarray(x, dtype=np.float64), tn.asarray(y, dtype=np.float64)))
            np.testing.assert_allclose(tl, gold["loss"][s], rtol=1e-9)
    finally:
        tn.set_default_float(np.float32)


@pytest.mark.gpu
def test_bucketed_overlapped_allreduce_world1(comm):
    """Large arena (8.4 MB): the sharded step all-reduces each layer's gradients on the communication stream right
    behind that layer's backward (tnn_allreduce_async / tnn_comm_join).  At world 1 every collective is the identity,
    so losses and parameters must equal the unsharded trainer's bit for bit — eagerly and replayed from a hipGraph
    (the event edges become a side branch of the graph)."""
    from tinynn_autograd_amd.fused import MLPTrainer
    rs = np.random.RandomState(9)
    widths, m = [1024, 1024, 1024], 256
    a = np.sqrt(6.0 / 2048)
    layers = [{"w": rs.uniform(-a, a, (1024, 1024)).astype(np.float32), "b": np.zeros((1, 1024), np.float32)}
              for _ in range(2)]
    data = [tn.asarray(rs.rand(m, 1024).astype(np.float32)) for _ in range(4)]
    results = []
    for c in (None, comm):
        t = MLPTrainer(widths, m, loss="mse", optimizer="adam", lr=1e-3, comm=c, force_dp=c is not None)
        t.set_parameters(layers)
        losses = [float(t.step(x, x)) for x in data[:2]]
        graph = t.capture_steps([(x, x) for x in data[2:]])
        losses += [float(v) for v in np.asarray(graph.launch())]
        results.append((losses, np.asarray(t.params)))
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])
