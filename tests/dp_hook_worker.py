"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N run of the product's OWN C++ data-parallel step
(tnn_mlp_step_sharded in csrc/tnn_mlp.cpp: shard statistics exchange, per-layer gradient buckets, Adam in bucket order,
the loss riding in the last bucket) on the CPU twin, whose collectives are routed to torch.distributed/gloo through
host callbacks (tnn_twin_set_collectives — the twin's "device" memory is host memory).  What runs on the MI355X through
RCCL runs here through gloo; the launch sequence and the host logic are the same code.

  mode "D"        strong-scaling config D: global batch 1024, 1024/W rows per rank, softmax loss, against
                  the reference's single-process trajectory tests/golden/traj_D_adam.npz — all-reduce + Adam tail
  mode "Dbucket"  the same with the arena split into one bucket per layer (TNN_BUCKET_BYTES=1)
  mode "Cbucket"  config-C-small (256-256-256 autoencoder, bs 64, sum-of-squares), bucketed, against traj_C_small.npz
  mode "A"        the bs-128 trajectory split over the ranks (32 rows each at world 4): the shard's head fits the one-launch
                  form, so the step takes the RCCL-shaped 6-launch structure — statistics as a launch of their own
                  (tnn_mlp_head_stats), ONE all-gather of the pairs, head + hidden backward merging them
                  (tnn_mlp_head_bwd_tick_ext), ONE all-reduce with the Adam tail — against traj_A_adam.npz
"""

import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_NP = {0: np.float32, 1: np.float64, 2: np.int64, 3: np.uint8}


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
    calls = {"allreduce": [], "allgather": 0}

    def as_array(ptr, n, dtype):
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n * np.dtype(_NP[dtype]).itemsize,)).view(_NP[dtype])

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int)
    def allreduce(buf, n, dtype, rop):
        arr = as_array(buf, n, dtype)
        t = torch.from_numpy(arr)                                   # shares the twin's memory: reduced in place
        dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[rop])
        calls["allreduce"].append(int(n))
        return 0

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int)
    def allgather(send, recv, n, dtype):
        src = torch.from_numpy(as_array(send, n, dtype).copy())
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        as_array(recv, n * world, dtype)[...] = np.concatenate([p.numpy() for p in parts])
        calls["allgather"] += 1
        return 0

    hook = lib.cdll.tnn_twin_set_collectives
    hook.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    hook.restype = ctypes.c_int
    assert hook(rank, world, ctypes.cast(allreduce, ctypes.c_void_p), ctypes.cast(allgather, ctypes.c_void_p)) == 0
    comm = DeviceCommunicator(rank, world)           # the product's communicator class over tnn_allreduce / tnn_allgather

    name = "C_small" if mode == "Cbucket" else "A_adam" if mode == "A" else "D_adam"
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
            sizes = [w[l] * w[l + 1] + w[l + 1] for l in range(n_layers)]
            want = [sizes[-1] + 1] + sizes[-2::-1]
            assert calls["allreduce"] == want, (calls["allreduce"], want)
        else:
            assert calls["allreduce"] == [n_params + 1], calls["allreduce"]       # ONE collective: arena + loss slot
        if s == 0 and cfg["loss"] == "softmax_nll":
            assert calls["allgather"] == 1                                         # the {max, sum-exp} exchange (C2)
    np.testing.assert_allclose(losses, gold["loss"], rtol=1e-5, err_msg="%s sharded over %d ranks" % (name, world))
    for l in range(n_layers):
        for k in ("w", "b"):
            H.check_summary(np.asarray(trainer.param_view(l, k)), gold, "final_%d%s" % (l, k), rtol=0, atol=0.1 * cfg["lr"])
    flat = np.asarray(trainer.params)
    parts = [None] * world
    dist.all_gather_object(parts, flat.tobytes())
    assert all(p == parts[0] for p in parts), "parameters diverged across ranks"
    dist.barrier()
    hook(0, 1, None, None)
    print("dp_hook_worker %s rank %d/%d ok" % (mode, rank, world))


if __name__ == "__main__":
    main()
