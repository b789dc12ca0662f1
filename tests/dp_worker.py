"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N data-parallel run on the CPU twin with
the gloo communicator.  Checks that sharding the bs=1024 batch over the ranks reproduces the reference's
single-process trajectory (tests/golden/traj_D_adam.npz): whole-batch softmax exchange (C2), gradient
all-reduce (C1), identical parameters on every rank."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import conftest
    from tinynn_autograd_amd import _lib
    _lib.install_test_twin(conftest.build_twin())
    import tinynn_autograd_amd as tn
    import helpers as H
    from tinynn_autograd_amd.core.tensor import Tensor
    comm = tn.dist.init_from_env(backend="gloo")
    rank, world = comm.rank, comm.world
    cfg, gold = H.load_traj("D_adam")
    w, m = cfg["widths"], cfg["m"]
    rows = m // world
    sl = slice(rank * rows, (rank + 1) * rows)
    steps = 3

    # ---- (a) op-level drop-in path: Model + fused loss with the communicator
    model, loss_layer = H.build_model(cfg, comm=comm)
    # ---- (b) whole-step trainer with the communicator (same initial parameters)
    trainer = tn.trainer_from_net(model.net, max_rows=rows, loss="softmax_nll", optimizer="adam", lr=cfg["lr"], comm=comm)
    for s, (x, y) in enumerate(H.batches(cfg["data_seed"], steps, m, w[0], w[-1], cfg["loss"])):
        model.zero_grad()
        pred = model.forward(Tensor(x[sl]))
        loss = loss_layer.loss(pred, Tensor(y[sl]))
        loss.backward()
        model.step()
        share = tn.asarray(np.array([float(loss.values)]))
        total = float(np.asarray(comm.allreduce(share))[0])
        np.testing.assert_allclose(total, gold["loss"][s], rtol=1e-5, err_msg="ops path loss step %d" % s)
        tl = float(trainer.step(tn.asarray(x[sl]), tn.asarray(y[sl])))        # loss slot is all-reduced
        np.testing.assert_allclose(tl, gold["loss"][s], rtol=1e-5, err_msg="trainer loss step %d" % s)
        if s == 0:
            z = np.asarray(pred.values, dtype=np.float64)
            full = np.concatenate([np.asarray(p) for p in np.asarray(comm.allgather(tn.asarray(z.astype(np.float32).ravel())))])
            H.check_summary(full.reshape(m, w[-1]), gold, "logits_0", rtol=0, atol=1e-5 * np.abs(full).max())
    # parameters identical on every rank and equal to the two paths' results
    flat = np.concatenate([np.asarray(l.params[k].values).ravel() for l in H.dense_layers(model) for k in ("w", "b")])
    both = np.asarray(comm.allgather(tn.asarray(flat)))
    assert np.array_equal(both[0], both[-1]), "parameters diverged across ranks"
    np.testing.assert_allclose(np.asarray(trainer.params), flat, rtol=0, atol=2e-5)
    comm.barrier()
    print("dp_worker rank %d/%d ok" % (rank, world))


if __name__ == "__main__":
    main()
