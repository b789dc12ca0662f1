"""The numpy oracle against the committed fixtures (which carry the REAL reference's outputs, written by
oracle/gen_golden.py in the build container).  Bit-exact for the op-graph restatement."""

import numpy as np
import pytest

import helpers as H
import op_cases
from oracle import ref_nn
from oracle.closed_form import ClosedFormMLP
from oracle.gen_golden import OracleOps
from oracle.ref_autograd import RefTensor


def test_op_cases_bit_exact():
    golden = H.load_op_cases()
    assert set(golden) == set(op_cases.CASES)
    for name, fn in op_cases.CASES.items():
        got = fn(RefTensor, OracleOps())
        for key, ref in golden[name].items():
            val = np.asarray(got[key], dtype=np.float64)
            assert val.shape == ref.shape and np.array_equal(val, ref), "%s/%s" % (name, key)


def _run_ref_nn(cfg):
    w = cfg["widths"]
    np.random.seed(cfg["seed"])
    layers = ref_nn.build_mlp(w)
    opt = ref_nn.Adam(lr=cfg["lr"]) if cfg["opt"] == "adam" else ref_nn.SGD(lr=cfg["lr"])
    loss_fn = ref_nn.softmax_nll if cfg["loss"] == "softmax_nll" else ref_nn.squared_error
    losses, logits = [], {}
    for s, (x, y) in enumerate(H.batches(cfg["data_seed"], cfg["steps"], cfg["m"], w[0], w[-1], cfg["loss"])):
        lv, pred = ref_nn.train_step(layers, opt, loss_fn, x, y)
        losses.append(float(lv))
        logits[s] = np.array(pred)
    return losses, logits, layers


@pytest.mark.parametrize("name", ["A_adam", "A_sgd", "A_ragged", "C_small", "R_example", "R_example_D"])
def test_trajectory_bit_exact(name):
    cfg, gold = H.load_traj(name)
    if name in ("A_adam", "A_sgd"):
        cfg = dict(cfg, steps=6)                      # keep the CPU suite short; prefix of the same run
    losses, logits, layers = _run_ref_nn(cfg)
    assert np.array_equal(np.array(losses), gold["loss"][:cfg["steps"]])
    for s in (0, 1):
        if "argmax_%d" % s in gold and s < cfg["steps"]:
            assert np.array_equal(np.argmax(logits[s], axis=1), gold["argmax_%d" % s])
            H.check_summary(logits[s], gold, "logits_%d" % s, rtol=0, atol=0)


def test_closed_form_matches_reference_trajectory():
    cfg, gold = H.load_traj("A_ragged")
    w = cfg["widths"]
    np.random.seed(cfg["seed"])
    layers = [l for l in ref_nn.build_mlp(w) if l.params]
    mlp = ClosedFormMLP([l.params["w"].values for l in layers], [l.params["b"].values for l in layers],
                        loss=cfg["loss"], optimizer=cfg["opt"], lr=cfg["lr"])
    for s, (x, y) in enumerate(H.batches(cfg["data_seed"], cfg["steps"], cfg["m"], w[0], w[-1], cfg["loss"])):
        loss, out, gW, gb = mlp.step(x, y)
        np.testing.assert_allclose(loss, gold["loss"][s], rtol=1e-6)
        if s == 0:
            H.check_summary(gW[2], gold, "grad0_2w", rtol=0, atol=2e-5 * np.abs(gold["grad0_2w"]).max())


def test_layers_and_epoch_loop_bit_exact():
    """ref_nn.Tanh and ref_nn.train_epochs against the reference's outputs (layers.npz, epoch.npz)."""
    import json
    import synth
    gold = dict(np.load(H.GOLDEN + "/layers.npz"))
    x, g = synth.layer_inputs()
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        t = RefTensor(x.astype(dt), requires_grad=True)
        y = ref_nn.Tanh().forward(t)
        y.backward(g)
        assert np.array_equal(np.asarray(y.values, dtype=np.float64), gold["tanh_%s_out" % tag])
        assert np.array_equal(t.grad, gold["tanh_%s_grad" % tag])
    # the formula is tanh(x / 2), not tanh(x)
    np.testing.assert_allclose(gold["tanh_f64_out"], np.tanh(x / 2), rtol=1e-12, atol=1e-15)
    ep = dict(np.load(H.GOLDEN + "/epoch.npz"))
    cfg = json.loads(str(ep["config"]))
    train_x, train_y, pool_x, pool_y = synth.epoch_dataset(cfg)
    test_x, test_y = pool_x[ep["test_rows"]], pool_y[ep["test_rows"]]
    np.random.seed(cfg["seed"])
    losses, preds, results = ref_nn.train_epochs(cfg["widths"], train_x, np.eye(10)[train_y], test_x, test_y, 1,
                                                 cfg["batch_size"], cfg["lr"])
    assert np.array_equal(np.array(losses), ep["loss"][:8])
    assert np.array_equal(preds[0], ep["argmax"][0]) and results[0]["hit_num"] == int(ep["hit_num"][0])
