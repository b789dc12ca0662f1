"""Container-only: the reference's UNMODIFIED core/layers.py, losses.py, optimizer.py, model.py, nn.py and
initializer.py run on this package's tensor/ops (north_star: "run unmodified as drop-ins").

The reference files are imported from /root/reference with `core.tensor` / `core.ops` pre-seeded in
sys.modules by this package's modules; nothing is copied.  Skipped where the reference is absent (GPU box).
"""

import os
import sys
import types

import numpy as np
import pytest

import helpers as H

REF = os.environ.get("TNN_REFERENCE_DIR", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "core")),
                                reason="reference checkout not present (it never travels to the GPU box)")


@pytest.fixture()
def ref_core():
    import tinynn_autograd_amd as tn
    saved = {k: v for k, v in sys.modules.items() if k == "core" or k.startswith("core.")}
    for k in saved:
        del sys.modules[k]
    stub = types.ModuleType("core")
    stub.__path__ = [os.path.join(REF, "core")]
    sys.modules["core"] = stub
    sys.modules["core.tensor"] = tn.core.tensor          # the seam: same names, device implementation
    sys.modules["core.ops"] = tn.core.ops
    old_flag = sys.dont_write_bytecode
    sys.dont_write_bytecode = True                       # never write into the read-only reference tree
    try:
        import core.layers, core.losses, core.optimizer, core.model, core.nn, core.initializer  # noqa
        yield sys.modules["core"]
    finally:
        sys.dont_write_bytecode = old_flag
        for k in [k for k in sys.modules if k == "core" or k.startswith("core.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_unmodified_reference_stack_trains_on_device(ref_core):
    import core.layers as L, core.losses as LO, core.optimizer as O, core.model as M, core.nn as NN
    from tinynn_autograd_amd.core.tensor import Tensor
    from tinynn_autograd_amd import device_array as da
    assert L.__file__.startswith(REF) and O.__file__.startswith(REF)
    cfg, gold = H.load_traj("A_adam")
    w = cfg["widths"]
    np.random.seed(cfg["seed"])
    layers = []
    for i in range(len(w) - 1):
        layers.append(L.Dense(w[i + 1], num_in=w[i]))
        if i < len(w) - 2:
            layers.append(L.ReLU())
    model = M.Model(net=NN.Net(layers), loss=LO.SoftmaxCrossEntropyLoss(), optimizer=O.Adam(lr=cfg["lr"]))
    loss_layer = LO.SoftmaxCrossEntropyLoss()
    for s, (x, y) in enumerate(H.batches(cfg["data_seed"], 6, cfg["m"], w[0], w[-1], cfg["loss"])):
        model.zero_grad()
        pred = model.forward(Tensor(x))
        loss = loss_layer.loss(pred, Tensor(y))
        loss.backward()
        model.step()
        assert isinstance(loss.values, da.DeviceArray) and isinstance(layers[0].params["w"].values, da.DeviceArray)
        np.testing.assert_allclose(float(loss.values), gold["loss"][s], rtol=1e-5)
        if s in (0, 1):
            H.check_summary(np.asarray(pred.values), gold, "logits_%d" % s, rtol=0,
                            atol=1e-5 * np.abs(np.asarray(pred.values)).max())
            assert np.array_equal(np.argmax(pred, axis=1), gold["argmax_%d" % s])
    # optimizer state stayed on the device: the reference's Adam ran as 13 DeviceArray expressions
    assert isinstance(model.optimizer._m, da.DeviceArray) and isinstance(model.optimizer._v, da.DeviceArray)


def test_unmodified_sigmoid_and_sgd(ref_core):
    import core.layers as L, core.optimizer as O
    from tinynn_autograd_amd.core.tensor import Tensor
    x = np.linspace(-4, 4, 12).reshape(3, 4)
    t = Tensor(x, requires_grad=True)
    s = L.Sigmoid().forward(t)                       # raises TypeError in the reference itself
    s.backward(np.ones((3, 4)))
    ref = 1 / (1 + np.exp(-x))
    np.testing.assert_allclose(np.asarray(s.values), ref, rtol=2e-6)
    np.testing.assert_allclose(np.asarray(t.grad), ref * (1 - ref), rtol=1e-5, atol=1e-7)
    g = Tensor(np.arange(6.0).reshape(2, 3)).values
    steps = O.SGD(lr=0.5).compute_step([{"w": g}], [{"w": Tensor(np.zeros((2, 3)))}])
    np.testing.assert_allclose(np.asarray(steps[0]["w"]), -0.5 * np.arange(6.0).reshape(2, 3))
