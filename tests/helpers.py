"""Shared helpers: golden fixtures, synthetic batches (same generator as oracle/gen_golden.py), model
builders on the device backend, comparison utilities."""

import json
import os

import numpy as np

import tinynn_autograd_amd as tn
from tinynn_autograd_amd.core.layers import Dense, ReLU
from tinynn_autograd_amd.core.losses import SoftmaxCrossEntropyLoss, SquaredErrorLoss
from tinynn_autograd_amd.core.model import Model
from tinynn_autograd_amd.core.nn import Net
from tinynn_autograd_amd.core.optimizer import SGD, Adam
from tinynn_autograd_amd.core.tensor import Tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_traj(name):
    data = dict(np.load(os.path.join(GOLDEN, "traj_%s.npz" % name)))
    cfg = json.loads(str(data.pop("config")))
    return cfg, data


def load_op_cases():
    with open(os.path.join(GOLDEN, "op_cases.json")) as f:
        raw = json.load(f)
    return {name: {k: np.array(v["data"], dtype=np.float64).reshape(v["shape"]) for k, v in res.items()}
            for name, res in raw.items()}


def batches(seed, steps, m, n_in, n_out, kind):
    """Identical to oracle/gen_golden.py:batches (the fixtures were produced on these inputs)."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(steps):
        x = rs.rand(m, n_in).astype(np.float32)
        if kind == "softmax_nll":
            # MNIST-like sparsity: ~19 % of the pixels are non-zero.  Dense uniform inputs make
            # the Adam trajectory chaotic (every step shifts all pre-activations coherently, ReLU units flip
            # for whole batches) and float32 vs float64 then diverges after ~8 steps for reasons unrelated to
            # kernel correctness; with this mask float32 tracks the float64 reference to ~4e-8 over 40 steps.
            x *= (rs.rand(m, n_in) < 0.19)
        y = np.eye(n_out)[rs.randint(0, n_out, m)] if kind == "softmax_nll" else x.copy()
        out.append((x, y))
    return out


def build_model(cfg, fused=True, comm=None, use_arena=True):
    """Dense/ReLU net, eager init under np.random.seed(cfg.seed) in layer order."""
    w = cfg["widths"]
    np.random.seed(cfg["seed"])
    layers = []
    for i in range(len(w) - 1):
        layers.append(Dense(w[i + 1], num_in=w[i], fused=fused))
        if i < len(w) - 2:
            layers.append(ReLU())
    net = Net(layers)
    if cfg["opt"] == "adam":
        opt = Adam(lr=cfg["lr"], fused=fused)
    else:
        opt = SGD(lr=cfg["lr"])
    loss = SoftmaxCrossEntropyLoss(fused=fused, comm=comm) if cfg["loss"] == "softmax_nll" else SquaredErrorLoss()
    return Model(net=net, loss=loss, optimizer=opt, comm=comm, use_arena=use_arena), loss


def dense_layers(model):
    return [l for l in model.net.layers if isinstance(l, Dense)]


def check_summary(arr, data, prefix, rtol, atol):
    """Compare `arr` with a fixture entry stored either in full or as (sum, abs_sum, l2, sample@idx)."""
    arr = np.asarray(arr, dtype=np.float64)
    if prefix in data:
        ref = data[prefix]
        assert arr.shape == ref.shape, "%s: shape %s vs %s" % (prefix, arr.shape, ref.shape)
        np.testing.assert_allclose(arr, ref, rtol=rtol, atol=atol, err_msg=prefix)
        return float(np.abs(arr - ref).max())
    flat = arr.ravel()
    idx = data[prefix + "_idx"]
    np.testing.assert_allclose(flat[idx], data[prefix + "_sample"], rtol=rtol, atol=atol, err_msg=prefix)
    np.testing.assert_allclose(np.sqrt((flat ** 2).sum()), data[prefix + "_l2"], rtol=max(rtol, 1e-5), err_msg=prefix + " l2")
    np.testing.assert_allclose(np.abs(flat).sum(), data[prefix + "_abs_sum"], rtol=max(rtol, 1e-5), err_msg=prefix + " abs_sum")
    return float(np.abs(flat[idx] - data[prefix + "_sample"]).max())


def max_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
