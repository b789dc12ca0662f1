"""MNIST-style MLP training driver — this build's counterpart of examples/mnist/run.py:45-93.

Same loop (zero_grad -> forward -> loss -> backward -> step per batch, per-epoch shuffle on numpy's global RNG,
evaluation = argmax over the logits -> AccEvaluator) and the same flags, on device Tensors.  No MNIST file exists
in this environment (no network), so `--data_dir` is accepted but the data are synthetic unless a
`mnist.pkl.gz` is found there.

    python -m tinynn_autograd_amd.examples.mnist_run --num_ep 2 --batch_size 128 --seed 0 [--trainer]

--trainer    run the epoch through the whole-step trainer (one hipGraph per epoch) instead of the op-level path
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
from tinynn_autograd_amd.core.evaluator import AccEvaluator                  # noqa: E402
from tinynn_autograd_amd.core.layers import Dense, ReLU                      # noqa: E402
from tinynn_autograd_amd.core.losses import SoftmaxCrossEntropyLoss         # noqa: E402
from tinynn_autograd_amd.core.model import Model                             # noqa: E402
from tinynn_autograd_amd.core.nn import Net                                  # noqa: E402
from tinynn_autograd_amd.core.optimizer import Adam                          # noqa: E402
from tinynn_autograd_amd.core.tensor import Tensor                           # noqa: E402
from tinynn_autograd_amd.utils.data_iterator import BatchIterator            # noqa: E402
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


def train(train_x, train_y, test_x, test_y, widths, num_ep, batch_size, lr, trainer=False, log=None):
    """The loop of examples/mnist/run.py:50-93 on device Tensors.  `train_y` / `test_y` are integer labels.  Returns
    (loss_list as floats, per-epoch argmax vectors, per-epoch AccEvaluator dicts).

    RNG order is the reference's: the first epoch's shuffle is drawn BEFORE the lazily initialised Dense layers draw
    their weights at the first forward (core/layers.py:45-46) — also on the trainer path."""
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
    iterator = BatchIterator(batch_size=batch_size)
    evaluator = AccEvaluator()
    step_trainer = None
    loss_list, preds, results = [], [], []
    for epoch in range(num_ep):
        t_start = time.time()
        if trainer:
            batches = [(b.inputs.values, b.targets.values) for b in iterator(train_x, train_y)]   # shuffle drawn here
            if step_trainer is None:
                model.forward(Tensor(batches[0][0][:1]))   # lazy Dense init from the first batch's width
                step_trainer = tn.trainer_from_net(net, max_rows=batch_size, lr=lr, dtype=tn.get_default_float())
            full = [b for b in batches if b[0].shape[0] == batch_size]
            graph = step_trainer.capture_steps(full)       # the epoch's full batches as ONE hipGraph launch
            loss_list.extend(float(v) for v in np.asarray(graph.launch()))
            for x, y in batches[len(full):]:               # ragged last batch
                loss_list.append(float(step_trainer.step(x, y)))
            for i, layer in enumerate(l for l in net.layers if isinstance(l, Dense)):
                layer.params["w"].values = step_trainer.param_view(i, "w")
                layer.params["b"].values = step_trainer.param_view(i, "b")
        else:
            device_losses = []
            for batch in iterator(train_x, train_y):
                model.zero_grad()
                pred = model.forward(batch.inputs)
                loss = loss_layer.loss(pred, batch.targets)
                loss.backward()
                model.step()
                device_losses.append(loss.values)          # a 0-d DeviceArray: no host sync inside the loop
            loss_list.extend(float(v) for v in device_losses)
        tn.synchronize()
        if log:
            log("Epoch %d tim cost: %.4f" % (epoch, time.time() - t_start))
        model.set_phase("TEST")
        test_pred = model.forward(test_x)
        test_pred_idx = np.argmax(test_pred, axis=1)
        res = evaluator.evaluate(test_pred_idx, np.asarray(test_y))
        preds.append(test_pred_idx)
        results.append(res)
        if log:
            log(res)
        model.set_phase("TRAIN")
    return loss_list, preds, results


def main(args):
    if args.seed >= 0:
        random_seed(args.seed)
    (train_x, train_y), (test_x, test_y), source = prepare_dataset(args.data_dir)
    print("data: %s, %d train / %d test rows" % (source, len(train_x), len(test_x)))
    print("backend:", tn.backend_name())
    widths = [int(w) for w in args.widths.split(",")]
    loss_list, _, _ = train(train_x, train_y, test_x, test_y, widths, args.num_ep, args.batch_size, args.lr,
                            trainer=args.trainer, log=print)
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
    main(parser.parse_args())
