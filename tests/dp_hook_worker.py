"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N run of the product's OWN C++ data-parallel step
(tnn_mlp_step_sharded in csrc/tnn_mlp.cpp: shard statistics exchange, per-layer gradient buckets, Adam in bucket order,
the loss riding in the last bucket) on the CPU twin, whose collectives are routed to torch.distributed/gloo through
host callbacks (tnn_twin_set_collectives — the twin's "device" memory is host memory).  What runs on the MI355X through
RCCL runs here through gloo; the launch sequence and the host logic are the same code.

  mode "D"        strong-scaling config D: global batch 1024, 1024/W rows per rank, softmax loss, against
                  the reference's single-process trajectory tests/golden/traj_D_adam.npz — all-reduce + Adam tail
                  (256 rows per rank at world 4: the row-blocked form of the one-launch head, two blocks of 128 rows)
  mode "Rbucket"  the reference's OWN example net (784-200-100-70-30-10, tests/golden/traj_R_example.npz) split over the ranks:
                  no fast head exists for its 70 -> 30 -> 10 tail, so the step takes the generic path — here with the arena
                  split into one gradient bucket per layer (TNN_BUCKET_BYTES=1)
  mode "Cbucket"  config-C-small (256-256-256 autoencoder, bs 64, sum-of-squares), bucketed, against traj_C_small.npz
  mode "A"        the bs-128 trajectory split over the ranks (32 rows each at world 4): the shard's head fits the one-launch
                  form, so the step takes the RCCL-shaped 6-launch structure — statistics as a launch of their own
                  (tnn_mlp_head_stats), ONE all-gather of the pairs, head + hidden backward merging them
                  (tnn_mlp_head_bwd_tick_ext), ONE all-reduce with the Adam tail — against traj_A_adam.npz
  mode "E"        configs[4] in small: bf16 trainer (512-wide x 2 layers, sum-of-squares, Adam), 64 rows per rank — the
                  SHARDED-OPTIMIZER step (mlp16_step_zero): per layer reduce-scatter of the bf16 weight gradient, Adam on
                  the owned rows, all-gather of the bf16 rows; one small all-reduce for biases + loss.  Against
                  oracle/closed_form.py (float64) on the same bf16-rounded weights and inputs, collective order and sizes
                  asserted, bf16 weights identical on every rank.
"""

import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_NP = {0: np.float32, 1: np.float64, 2: np.int64, 3: np.uint8, 4: np.uint16}   # 4: bf16 bit patterns


def main():
    mode = sys.argv[1]
    import torch
    import torch.distributed as dist
    import conftest
    from tinynn_autograd_amd import _lib
    lib = _lib.install_test_twin(conftest.build_twin())
    import tinynn_autograd_amd as tn
    import helpers as H
    from tinynn_autograd_amd.dist import DeviceCommunicator
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    calls = {"allreduce": [], "allgather": 0, "seq": []}

    def as_array(ptr, n, dtype):
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n * np.dtype(_NP[dtype]).itemsize,)).view(_NP[dtype])

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int)
    def allreduce(buf, n, dtype, rop):
        arr = as_array(buf, n, dtype)
        t = torch.from_numpy(arr)                                   # shares the twin's memory: reduced in place
        dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[rop])
        calls["allreduce"].append(int(n))
        calls["seq"].append(("allreduce", int(n), int(dtype)))
        return 0

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int)
    def allgather(send, recv, n, dtype):
        src = torch.from_numpy(as_array(send, n, dtype).copy().view(np.uint8))       # bytes: gloo has no 16-bit integers
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        as_array(recv, n * world, dtype)[...] = np.concatenate([p.numpy() for p in parts]).view(_NP[dtype])
        calls["allgather"] += 1
        calls["seq"].append(("allgather", int(n), int(dtype)))
        return 0

    def bf16_round(f):
        u = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int)
    def reduce_scatter(send, recv, n, dtype):
        # every rank's whole send buffer travels (test infrastructure, not a bandwidth-optimal collective); slice `rank` is
        # summed in rank order, rounding to the wire dtype after every addition like a ring of bf16 adds would
        src = torch.from_numpy(as_array(send, n * world, dtype).copy().view(np.uint8))
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        mine = [p.numpy().view(_NP[dtype])[rank * n:(rank + 1) * n] for p in parts]
        if dtype == 4:
            acc = (mine[0].astype(np.uint32) << 16).view(np.float32)
            for q in mine[1:]:
                acc = (bf16_round(acc + (q.astype(np.uint32) << 16).view(np.float32)).astype(np.uint32) << 16).view(np.float32)
            out = bf16_round(acc)
        else:
            out = mine[0].copy()
            for q in mine[1:]:
                out = out + q
        as_array(recv, n, dtype)[...] = out
        calls["seq"].append(("reduce_scatter", int(n), int(dtype)))
        return 0

    hook = lib.cdll.tnn_twin_set_collectives
    hook.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    hook.restype = ctypes.c_int
    assert hook(rank, world, ctypes.cast(allreduce, ctypes.c_void_p), ctypes.cast(allgather, ctypes.c_void_p)) == 0
    hook_rs = lib.cdll.tnn_twin_set_reduce_scatter
    hook_rs.argtypes = [ctypes.c_void_p]
    hook_rs.restype = ctypes.c_int
    assert hook_rs(ctypes.cast(reduce_scatter, ctypes.c_void_p)) == 0
    comm = DeviceCommunicator(rank, world)           # the product's communicator class over tnn_allreduce / tnn_allgather
    if mode == "E":
        run_config_e_small(tn, comm, calls, rank, world, dist)
        dist.barrier()
        hook(0, 1, None, None)
        hook_rs(None)
        print("dp_hook_worker %s rank %d/%d ok" % (mode, rank, world))
        return

    name = {"Cbucket": "C_small", "A": "A_adam", "Rbucket": "R_example"}.get(mode, "D_adam")
    cfg, gold = H.load_traj(name)
    w, m = cfg["widths"], cfg["m"]
    assert m % world == 0
    rows = m // world
    sl = slice(rank * rows, (rank + 1) * rows)
    model, _ = H.build_model(cfg)
    trainer = tn.trainer_from_net(model.net, max_rows=rows, loss=cfg["loss"], optimizer="adam", lr=cfg["lr"], comm=comm)
    n_layers, n_params = trainer.n_layers, trainer.n_params
    losses = []
    for s, (x, y) in enumerate(H.batches(cfg["data_seed"], cfg["steps"], m, w[0], w[-1], cfg["loss"])):
        calls["allreduce"] = []
        losses.append(float(trainer.step(tn.asarray(x[sl]), tn.asarray(y[sl]))))
        if mode.endswith("bucket"):
            # one all-reduce per layer, last layer first, the last layer's bucket carrying the loss slot
            pw = trainer._pwidths                         # (hidden widths padded to multiples of 16 inside the arenas)
            sizes = [pw[l] * pw[l + 1] + pw[l + 1] for l in range(n_layers)]
            want = [sizes[-1] + 1] + sizes[-2::-1]
            assert calls["allreduce"] == want, (calls["allreduce"], want)
        else:
            assert calls["allreduce"] == [n_params + 1], calls["allreduce"]       # ONE collective: arena + loss slot
        if s == 0 and cfg["loss"] == "softmax_nll":
            assert calls["allgather"] == 1                                         # the {max, sum-exp} exchange (C2)
    np.testing.assert_allclose(losses, gold["loss"], rtol=1e-5, err_msg="%s sharded over %d ranks" % (name, world))
    for l in range(n_layers):
        for k in ("w", "b"):
            H.check_summary(np.asarray(trainer.param_view(l, k)), gold, "final_%d%s" % (l, k), rtol=0, atol=0.01 * cfg["lr"])       # (parity_suite.ADAM_GATE)
    flat = np.asarray(trainer.params)
    parts = [None] * world
    dist.all_gather_object(parts, flat.tobytes())
    assert all(p == parts[0] for p in parts), "parameters diverged across ranks"
    dist.barrier()
    hook(0, 1, None, None)
    hook_rs(None)
    print("dp_hook_worker %s rank %d/%d ok" % (mode, rank, world))


def run_config_e_small(tn, comm, calls, rank, world, dist, width=512, loss_rtol=2e-2, plain_rtol=None):
    """calls: the collective log of the twin's hooks, or None on the HIP library (tests/p2p_zero_worker.py: the same checks with
    the collectives on the peer-to-peer transport's bulk path, several ranks on one GPU)."""
    from oracle.closed_form import ClosedFormMLP                 # the checker
    from tinynn_autograd_amd import bf16
    from tinynn_autograd_amd.fused import MLPTrainer
    BF16, F32 = 4, 0
    widths, rows, lr, steps = [width, width, width], 64, 1e-3, 3
    m = rows * world
    rs = np.random.RandomState(91)
    a = np.sqrt(6.0 / (2 * width))
    W = [bf16.round_to_bf16(rs.uniform(-a, a, (width, width)).astype(np.float32)) for _ in range(2)]
    B = [bf16.round_to_bf16((rs.randn(1, width) * 0.02).astype(np.float32)) for _ in range(2)]
    x = bf16.round_to_bf16(rs.rand(m, width).astype(np.float32))
    if calls is None:
        calls = {"seq": None}
    sl = slice(rank * rows, (rank + 1) * rows)
    trainer = MLPTrainer(widths, rows, loss="mse", optimizer="adam", lr=lr, dtype="bfloat16", comm=comm)
    trainer.set_parameters([{"w": W[i], "b": B[i]} for i in range(2)])
    oracle = ClosedFormMLP(W, B, loss="mse", optimizer="adam", lr=lr)
    x16 = bf16.to_bf16(x[sl])
    losses, ref = [], []
    for s in range(steps):
        if calls["seq"] is not None:
            calls["seq"] = []
        losses.append(float(trainer.step(x16, x16)))
        ref.append(oracle.step(x, x)[0])
        # last layer first: reduce-scatter of the bf16 gradient slice, all-gather of the refreshed bf16 rows; then ONE
        # small fp32 all-reduce carrying both bias gradients and the loss
        shard = [widths[l] // world * widths[l + 1] for l in range(2)]
        want = [("reduce_scatter", shard[1], BF16), ("allgather", shard[1], BF16),
                ("reduce_scatter", shard[0], BF16), ("allgather", shard[0], BF16),
                ("allreduce", widths[1] + widths[2] + 1, F32)]
        assert calls["seq"] is None or calls["seq"] == want, (calls["seq"], want)
    # bf16 activations / dz / dW add ~2^-9 relative noise per tensor (same bar as tests/test_gpu_bf16.py)
    np.testing.assert_allclose(losses, ref, rtol=loss_rtol)
    assert losses[-1] < losses[0]
    if plain_rtol is not None:
        # ... and against the UNSHARDED bf16 trainer of the same library on the global batch (one process, fp32 gradient arena):
        # the same bf16 activations / dz, so what is left is the partial dW's rounding to bf16 before the sum — an offset or
        # ownership mistake in the sharded step (rows of the wrong rank updated) would show here at once
        plain = MLPTrainer(widths, m, loss="mse", optimizer="adam", lr=lr, dtype="bfloat16")
        plain.set_parameters([{"w": W[i], "b": B[i]} for i in range(2)])
        xg = bf16.to_bf16(x)
        plain_losses = [float(plain.step(xg, xg)) for _ in range(steps)]
        np.testing.assert_allclose(losses, plain_losses, rtol=plain_rtol)
        assert losses[0] == plain_losses[0] or abs(losses[0] - plain_losses[0]) <= 1e-5 * plain_losses[0]      # same forward, same weights
    w16 = np.asarray(trainer.weights_bf16()).copy()
    parts = [None] * world
    dist.all_gather_object(parts, w16.tobytes())
    assert all(p == parts[0] for p in parts), "bf16 weights diverged across ranks"
    for l in range(2):
        got = (np.asarray(trainer.weights_bf16(l)).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        # Adam moves a weight by <= lr per step whatever the gradient's size: three steps of sign-like updates on gradients
        # that carry bf16 noise stay within a fraction of lr of the float64 trajectory, plus half a bf16 ulp (|w| < 2^-3)
        err = np.abs(got - oracle.W[l])
        if rank == 0:
            print("E layer %d: max %.3g  p99 %.3g  p90 %.3g  median %.3g" % (l, err.max(), np.quantile(err, 0.99), np.quantile(err, 0.9), np.median(err)))
        # (where a gradient element is smaller than its bf16 noise the update's sign is arbitrary: the device and the oracle
        # may then move up to lr per step in opposite directions)
        assert err.max() <= 2 * steps * lr + 2.0 ** -12, (l, err.max())
        assert np.quantile(err, 0.99) <= 1.0 * lr + 2.0 ** -12, (l, float(np.quantile(err, 0.99)))
        assert np.median(err) <= 0.1 * lr + 2.0 ** -12, (l, float(np.median(err)))
        # this rank's fp32 master rows are what its bf16 rows were rounded from
        r0, r1 = rank * widths[l] // world, (rank + 1) * widths[l] // world
        own = np.asarray(trainer._view(l, "w"))[r0:r1]
        assert np.array_equal(bf16.round_to_bf16(own), (np.asarray(trainer.weights_bf16(l))[r0:r1].astype(np.uint32) << 16).view(np.float32))
        bias = np.asarray(trainer.param_view(l, "b"), dtype=np.float64)
        assert np.abs(bias - oracle.b[l]).max() <= 2 * steps * lr and np.median(np.abs(bias - oracle.b[l])) <= 0.1 * lr
    # ---- the fp32 masters are sharded: readers of the whole matrix are refused until the owned slices have been gathered,
    # and a checkpoint (a collective) holds WHOLE, rank-identical arenas from which a fresh trainer continues bit for bit
    if world > 1:
        assert trainer.masters_sharded() == world
        # ... the moment arenas included, and the implicit collectives (state_dict / save / get_parameters) refuse to start
        # unless told collective=True: `if rank == 0: trainer.save(path)` must raise, not hang in the all-gather
        for reader in (lambda: trainer.param_view(0, "w"), lambda: trainer.param_view(0, "w", arena=trainer.arena_m),
                       lambda: trainer.param_view(1, "w", arena=trainer.arena_v), trainer.state_dict, trainer.get_parameters,
                       lambda: trainer.save("/tmp/never_written.npz")):
            try:
                reader()
                raise AssertionError("a reader of sharded masters did not raise")
            except RuntimeError:
                pass
        assert trainer.grad_view(0, "w").shape == (widths[0], widths[1])          # the gradient arena is not sharded
    if calls["seq"] is not None:
        calls["seq"] = []
    state = trainer.state_dict(collective=True)
    if world > 1 and calls["seq"] is not None:
        shard = [widths[l] // world * widths[l + 1] for l in range(2)]
        assert calls["seq"] == [("allgather", shard[l], F32) for l in range(2) for _ in range(3)], calls["seq"]
    assert trainer.masters_sharded() == 0
    parts = [None] * world
    dist.all_gather_object(parts, state["params"].tobytes() + state["m"].tobytes() + state["v"].tobytes())
    assert all(p == parts[0] for p in parts), "gathered fp32 arenas differ across ranks"
    for l in range(2):
        whole = np.asarray(trainer.param_view(l, "w"))
        assert np.array_equal(bf16.round_to_bf16(whole), (np.asarray(trainer.weights_bf16(l)).astype(np.uint32) << 16).view(np.float32))
    resumed = MLPTrainer(widths, rows, loss="mse", optimizer="adam", lr=lr, dtype="bfloat16", comm=comm)
    resumed.load_state_dict(state)
    la, lb = float(trainer.step(x16, x16)), float(resumed.step(x16, x16))
    assert la == lb, (la, lb)
    assert np.array_equal(np.asarray(trainer.weights_bf16()), np.asarray(resumed.weights_bf16()))
    sa, sb = trainer.state_dict(collective=True), resumed.state_dict(collective=True)
    for k in ("params", "m", "v", "pows"):
        assert np.array_equal(sa[k], sb[k]), k


if __name__ == "__main__":
    main()
