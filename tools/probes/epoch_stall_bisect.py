#!/usr/bin/env python3
"""Which INGREDIENT of the reference's loop (trainer path of examples/mnist_run.train) carries the one-off pause?  The same loop,
rebuilt here with switches (VARIANT, '+'-separated):
    base        everything as mnist_run.train does it
    noeval      no evaluation pass (forward on 10,000 test rows through the op-level API, argmax, read-back) after the epochs
    nogather    no per-epoch shuffle / device gather: the batches are fixed slices of the resident training set
    smallgraph  the epoch as seven launches of 64-step graphs (one graph per 64 batches) instead of ONE 391-step graph
    nolosses    no read-back of the 391 losses: a stream synchronisation instead
    fusedrun    the trainer, its data and its graphs replaced by bench.FusedRun's (the headline measurement's 64-step graph over
                synthetic batches, six launches per "epoch"); the dataset is still uploaded and the op-level forward still runs
    nolazy      (with fusedrun) no op-level forward of one row
    nodata      (with fusedrun) the training / test sets are not uploaded at all
prints capture / per-epoch step times of two runs of four epochs."""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np   # noqa: E402
import torch         # noqa: E402
import tinynn_autograd_amd as tn                         # noqa: E402
from tinynn_autograd_amd import _lib                     # noqa: E402
from tinynn_autograd_amd.examples import mnist_run as mr   # noqa: E402

torch.cuda.set_device(0)
lib = _lib.get()
V = set(os.environ.get("VARIANT", "base").split("+"))
(train_x, train_y), (test_x, test_y), source = mr.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)


def train(num_ep=4, batch_size=128, lr=1e-3):
    if "nodata" not in V:
        ty = mr.get_one_hot(train_y, 10)
        tx, ty = mr.Tensor(train_x), mr.Tensor(ty)
        ex = mr.Tensor(test_x)
    net = mr.Net([mr.Dense(256), mr.ReLU(), mr.Dense(128), mr.ReLU(), mr.Dense(10)])
    model = mr.Model(net=net, loss=mr.SoftmaxCrossEntropyLoss(), optimizer=mr.Adam(lr=lr))
    iterator = mr.BatchIterator(batch_size=batch_size, reuse_buffers=True)
    evaluator = mr.AccEvaluator()
    step_trainer, graphs, out = None, None, []
    fixed = None
    for epoch in range(num_ep):
        tn.synchronize()
        t_start = time.time()
        if "fusedrun" in V:
            pairs = None
        elif "nogather" in V:
            if fixed is None:
                n = len(train_x) // batch_size
                fixed = [(tx.values[i * batch_size:(i + 1) * batch_size], ty.values[i * batch_size:(i + 1) * batch_size]) for i in range(n)]
            pairs = fixed
        else:
            batches = list(iterator(tx, ty))
            pairs = [(b.inputs.values, b.targets.values) for b in batches]
        t_data = t_capture = time.time()
        if step_trainer is None and "fusedrun" in V:
            import bench
            if "nolazy" not in V:
                model.forward(mr.Tensor(np.zeros((1, 784), np.float32)))
            step_trainer = bench.FusedRun(bench.WIDTHS_A, 128, "softmax_nll", 64)
            graphs = [step_trainer.chunk] * 6
            t_capture = time.time()
        elif step_trainer is None:
            model.forward(mr.Tensor(pairs[0][0][:1]))
            step_trainer = tn.trainer_from_net(net, max_rows=batch_size, lr=lr, dtype=tn.get_default_float())
        if graphs is None:
            if "smallgraph" in V:
                graphs = [step_trainer.capture_steps(pairs[i:i + 64]) for i in range(0, len(pairs), 64)]
            else:
                graphs = [step_trainer.capture_steps(pairs)]
            t_capture = time.time()
        dev = [g.launch() for g in graphs]
        if epoch + 1 < num_ep and "nogather" not in V and "fusedrun" not in V:
            iterator.prefetch_order(len(train_x))
        if "nolosses" in V or "fusedrun" in V:
            lib.stream_sync()
        else:
            losses = [np.asarray(d) for d in dev]
        tn.synchronize()
        t_train = time.time()
        if "noeval" not in V and "fusedrun" not in V:
            model.set_phase("TEST")
            for i, layer in enumerate(l for l in net.layers if isinstance(l, mr.Dense)):
                layer.params["w"].values = step_trainer.param_view(i, "w")
                layer.params["b"].values = step_trainer.param_view(i, "b")
            pred = model.forward(ex)
            idx = np.argmax(pred, axis=1)
            evaluator.evaluate(idx, np.asarray(test_y))
            model.set_phase("TRAIN")
        out.append((t_capture - t_data, t_train - t_capture))
    return out


pre = ""
if os.environ.get("PREWARM", "0") != "0":
    # a process that has already worked for a while (bench.py's situation): the headline step's graph replayed for PREWARM ms
    import bench
    hold = bench.FusedRun(bench.WIDTHS_A, 128, "softmax_nll", 64)
    lib.stream_sync()
    t_end, n, longest = time.time() + float(os.environ["PREWARM"]) * 1e-3, 0, 0.0
    while time.time() < t_end:
        t = time.time()
        hold.chunk.launch()
        lib.stream_sync()
        longest = max(longest, time.time() - t)
        n += 1
    pre = "prewarm: %d replays, longest %.2f ms | " % (n, longest * 1e3)
    if os.environ.get("RELEASE", "0") == "1":
        # what bench.py does in front of its epoch loop: the earlier measurement's trainer, graphs and arrays are destroyed
        import gc
        del hold
        gc.collect()
        lib.stream_sync()
        pre += "released | "
res = []
for rep in range(2):
    np.random.seed(0)
    o = train()
    lib.stream_sync()
    res.append("capture %5.1f steps %s" % (o[0][0] * 1e3, " ".join("%6.2f" % (s * 1e3) for _, s in o)))
    paused = any(s > 0.02 for _, s in o) or o[0][0] > 0.04
    res[-1] += "  <-- paused" if paused else ""
print("VARIANT=%-28s | %srun 0: %s | run 1: %s" % ("+".join(sorted(V)), pre, res[0], res[1]))
