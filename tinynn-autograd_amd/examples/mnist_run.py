"""MNIST-style MLP training driver — this build's counterpart of examples/mnist/run.py:45-93.

Same loop (zero_grad -> forward -> loss -> backward -> step per batch, per-epoch shuffle on numpy's global RNG,
evaluation = argmax over the logits -> AccEvaluator) and the same flags, on device Tensors.  No MNIST file exists
in this environment (no network), so `--data_dir` is accepted but the data are synthetic unless a
`mnist.pkl.gz` is found there.

    python -m tinynn_autograd_amd.examples.mnist_run --num_ep 2 --batch_size 128 --seed 0 [--trainer]

--trainer    run the epoch through the whole-step trainer (ONE hipGraph of the epoch's steps, captured in epoch 0 and
             replayed every epoch) instead of the op-level path
--capture    op-level path recorded once with tn.capture (epoch 0 eager, epoch 1 recorded) and replayed afterwards
--widths     hidden widths (default 256,128 = BASELINE.json's 784-256-128-10; the reference example hard-codes
             200,100,70,30, examples/mnist/run.py:59-69)
"""

import argparse
import gzip
import os
import pickle
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))

import tinynn_autograd_amd as tn                                             # noqa: E402
from tinynn_autograd_amd import device_array as da                           # noqa: E402
from tinynn_autograd_amd.core.evaluator import AccEvaluator                  # noqa: E402
from tinynn_autograd_amd.core.layers import Dense, ReLU                      # noqa: E402
from tinynn_autograd_amd.core.losses import SoftmaxCrossEntropyLoss         # noqa: E402
from tinynn_autograd_amd.core.model import Model                             # noqa: E402
from tinynn_autograd_amd.core.nn import Net                                  # noqa: E402
from tinynn_autograd_amd.core.optimizer import Adam                          # noqa: E402
from tinynn_autograd_amd.core.tensor import Tensor                           # noqa: E402
from tinynn_autograd_amd.utils.data_iterator import BatchIterator            # noqa: E402
from tinynn_autograd_amd.utils.host_threads import fit_blas_pool_to_cpu_quota  # noqa: E402
from tinynn_autograd_amd.utils.seeder import random_seed                     # noqa: E402


def get_one_hot(targets, nb_classes):
    return np.eye(nb_classes)[np.array(targets).reshape(-1)]


def prepare_dataset(data_dir, n_train=51200, n_test=10000):
    path = os.path.join(data_dir, "mnist.pkl.gz")
    if os.path.exists(path):
        with gzip.open(path, "rb") as f:
            train_set, _, test_set = pickle.load(f, encoding="latin1")
        return train_set, test_set, "mnist.pkl.gz"
    rs = np.random.RandomState(1234)                       # SURVEY §8d synthetic recipe (MNIST-like 19 % sparsity)
    teacher = rs.randn(784, 10)                            # labels a network can learn: argmax of a fixed linear map

    def make(n):
        x = (rs.rand(n, 784) * (rs.rand(n, 784) < 0.19)).astype(np.float32)
        return x, np.argmax(x @ teacher, axis=1)
    return make(n_train), make(n_test), "synthetic (MNIST-like sparsity, linear-teacher labels)"


def train(train_x, train_y, test_x, test_y, widths, num_ep, batch_size, lr, trainer=False, capture=False, log=None,
          stats=None, reuse_buffers=True):
    """The loop of examples/mnist/run.py:50-93 on device Tensors.  `train_y` / `test_y` are integer labels.  Returns
    (loss_list as floats, per-epoch argmax vectors, per-epoch AccEvaluator dicts).

    Three ways through the epoch, same results (tests/parity_suite.py `epoch_loop_*`):
      default   the reference's loop body issued op by op from Python (core/tensor.py / core/ops.py seam)
      capture   the same loop body recorded ONCE with tn.capture (epoch 0 runs eagerly: lazy init, arena binding, optimizer
                state; epoch 1 is recorded) and replayed for every later epoch
      trainer   the whole-step trainer (fused.MLPTrainer), every step of the epoch in ONE hipGraph captured in epoch 0
    The replays work because `BatchIterator(reuse_buffers=True)` gathers every epoch's permutation into the same HBM
    buffers: the captured kernels read this epoch's rows at last epoch's addresses.  Per-step losses stay in HBM (one
    history vector) and come back with ONE device-to-host copy per epoch.

    RNG order is the reference's: the first epoch's shuffle is drawn BEFORE the lazily initialised Dense layers draw
    their weights at the first forward (core/layers.py:45-46) — also on the trainer path.

    stats: optional list; one dict per epoch is appended — wall-clock seconds of the phases `data` (shuffle + index upload
    + device gather), `capture` (graph capture + instantiation, 0.0 when a graph is replayed), `steps` (issue + run +
    the loss read-back), `train` (their sum = the reference's "Epoch .. tim cost"), `eval` (forward on the test rows +
    argmax + AccEvaluator), and `n_rows`."""
    train_y = get_one_hot(train_y, 10)
    train_x, train_y = Tensor(train_x), Tensor(train_y)    # resident in HBM for the whole run
    test_x = Tensor(test_x)
    layers = []
    for w in widths:
        layers += [Dense(w), ReLU()]
    layers.append(Dense(10))
    net = Net(layers)
    model = Model(net=net, loss=SoftmaxCrossEntropyLoss(), optimizer=Adam(lr=lr))
    loss_layer = SoftmaxCrossEntropyLoss()
    iterator = BatchIterator(batch_size=batch_size, reuse_buffers=reuse_buffers)
    evaluator = AccEvaluator()
    n_batches = iterator.num_batches(len(train_x))
    loss_hist = tn.empty((n_batches,), tn.get_default_float())   # this epoch's per-step losses, in HBM
    step_trainer = epoch_graph = epoch_key = loss_ptrs = None
    bound = False
    loss_list, preds, results = [], [], []

    def one_step(batch):
        model.zero_grad()
        pred = model.forward(batch.inputs)
        loss = loss_layer.loss(pred, batch.targets)
        loss.backward()
        model.step()
        return loss.values                                 # a 0-d DeviceArray: no host sync inside the loop

    for epoch in range(num_ep):
        tn.synchronize()
        t_start = time.time()
        batches = list(iterator(train_x, train_y))         # shuffle drawn here (or ahead, see below); ONE gather per array
        key = iterator.buffers_token                       # unchanged = the batches sit where the captured kernels read them
        more = epoch + 1 < num_ep
        t_data = time.time()
        t_capture = t_data
        t_sub = []
        if trainer:
            if step_trainer is None:
                model.forward(Tensor(batches[0].inputs.values[:1]))   # lazy Dense init from the first batch's width
                step_trainer = tn.trainer_from_net(net, max_rows=batch_size, lr=lr, dtype=tn.get_default_float())
            if epoch_graph is None or epoch_key != key:
                # every step of the epoch (the ragged last batch included) as ONE hipGraph launch
                epoch_graph = step_trainer.capture_steps([(b.inputs.values, b.targets.values) for b in batches])
                epoch_key = key
                t_capture = time.time()
            device_losses = epoch_graph.launch()           # asynchronous: the GPU works through the epoch ...
            t_sub.append(time.time())
            if more:
                iterator.prefetch_order(len(train_x))      # ... while the host draws the next epoch's permutation
            t_sub.append(time.time())
            losses = np.asarray(device_losses)             # the one synchronising read-back of the epoch
            t_sub.append(time.time())
            if not bound or step_trainer.padded:           # (views of the trainer's arena: bound once; padded nets: copies)
                for i, layer in enumerate(l for l in net.layers if isinstance(l, Dense)):
                    layer.params["w"].values = step_trainer.param_view(i, "w")
                    layer.params["b"].values = step_trainer.param_view(i, "b")
                bound = True
        elif capture and epoch > 0:
            if epoch_graph is None or epoch_key != key:
                epoch_graph = tn.capture(lambda: [one_step(b) for b in batches], warmup=0)
                epoch_key, loss_ptrs = key, None           # (the graph's loss buffers keep their addresses: listed once)
                t_capture = time.time()
            step_losses = epoch_graph()
            if more:
                iterator.prefetch_order(len(train_x))
            _, loss_ptrs = da.gather_scalars(step_losses, out=loss_hist, pointers=loss_ptrs)
            losses = np.asarray(loss_hist)
        else:
            step_losses = [one_step(batch) for batch in batches]
            da.gather_scalars(step_losses, out=loss_hist)  # the epoch's 391 losses: ONE launch + ONE device-to-host copy
            losses = np.asarray(loss_hist)
        loss_list.extend(losses.tolist())
        tn.synchronize()
        t_train = time.time()
        if log:
            log("Epoch %d tim cost: %.4f" % (epoch, t_train - t_start))
        model.set_phase("TEST")
        test_pred = model.forward(test_x)
        test_pred_idx = np.argmax(test_pred, axis=1)
        res = evaluator.evaluate(test_pred_idx, np.asarray(test_y))
        preds.append(test_pred_idx)
        results.append(res)
        if log:
            log(res)
        model.set_phase("TRAIN")
        if stats is not None:
            stats.append({"data": t_data - t_start, "capture": t_capture - t_data, "steps": t_train - t_capture,
                          "train": t_train - t_start, "eval": time.time() - t_train, "n_rows": len(train_x),
                          "n_steps": len(batches), "wall": (t_start, t_train),         # (time.time() at the epoch's start / end of training)
                          **({"launch": t_sub[0] - t_capture, "prefetch": t_sub[1] - t_sub[0], "readback": t_sub[2] - t_sub[1]}
                             if len(t_sub) == 3 else {})})
    return loss_list, preds, results


def main(args):
    if args.seed >= 0:
        random_seed(args.seed)
    pool = fit_blas_pool_to_cpu_quota()                    # (a 64-thread BLAS pool in a 16-CPU container stalls the loop: utils/host_threads.py)
    if pool["blas_threads"] != pool["blas_threads_before"]:
        print("host BLAS pool: %d -> %d threads (CPU quota %.1f)" % (pool["blas_threads_before"], pool["blas_threads"], pool["quota_cpus"]))
    (train_x, train_y), (test_x, test_y), source = prepare_dataset(args.data_dir)
    print("data: %s, %d train / %d test rows" % (source, len(train_x), len(test_x)))
    print("backend:", tn.backend_name())
    widths = [int(w) for w in args.widths.split(",")]
    loss_list, _, _ = train(train_x, train_y, test_x, test_y, widths, args.num_ep, args.batch_size, args.lr,
                            trainer=args.trainer, capture=args.capture, log=print)
    print("last loss: %.6f" % loss_list[-1])


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--num_ep", default=2, type=int)
    parser.add_argument("--data_dir", default="./examples/mnist/data", type=str)
    parser.add_argument("--lr", default=1e-3, type=float)
    parser.add_argument("--batch_size", default=128, type=int)
    parser.add_argument("--seed", default=-1, type=int)
    parser.add_argument("--widths", default="256,128", type=str)
    parser.add_argument("--trainer", action="store_true")
    parser.add_argument("--capture", action="store_true")
    main(parser.parse_args())
