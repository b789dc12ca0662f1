"""TEST INFRASTRUCTURE — container-only generator of tests/golden/* from the REAL reference.

    python oracle/gen_golden.py            (needs /root/reference; it is imported, never copied)

It (1) runs every case of tests/op_cases.py and a set of seeded training trajectories through the
reference's own core/ modules, (2) asserts that the numpy oracle (oracle/ref_autograd.py, ref_nn.py,
closed_form.py) reproduces them — bit-for-bit for the op-graph restatement, to float64 round-off for the
closed form — and (3) writes the reference's outputs as small fixtures.  The GPU box has no
/root/reference: there the fixtures ARE the reference.

Trajectory configs:  A = 784-256-128-10, bs 128, Adam 1e-3 and SGD 1e-2, 20 steps;
D = same net, bs 1024, Adam, 5 steps;  C-small = 256-256-256 autoencoder, bs 64, sum-of-squares/m, Adam,
5 steps;  eval = argmax + AccEvaluator on 1000 rows with forced ties;  layers = Tanh / ReLU objects forward + vjp and
a Dense-Tanh-Dense net;  epoch = the whole loop of examples/mnist/run.py (shuffle, ragged batch, 2 epochs, eval).
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("TNN_REFERENCE_DIR", "/root/reference")
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not found at %s — fixtures can only be generated in the build container" % REF)
    sys.path.insert(0, REF)
    import core.tensor as rt
    import core.ops as rops
    import core.layers as rlayers
    import core.losses as rlosses
    import core.optimizer as ropt
    import core.model as rmodel
    import core.nn as rnn
    import core.evaluator as reval
    return rt, rops, rlayers, rlosses, ropt, rmodel, rnn, reval


# ---------------------------------------------------------------------------- synthetic data
def batches(seed, steps, m, n_in, n_out, kind):
    """Deterministic synthetic batches (legacy RandomState => stable across numpy versions)."""
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
        if kind == "softmax_nll":
            y = np.eye(n_out)[rs.randint(0, n_out, m)]          # float64 one-hot, run.py:27-28
        else:
            y = x.copy()                                        # autoencoder target
        out.append((x, y))
    return out


CONFIGS = {
    "A_adam": dict(widths=[784, 256, 128, 10], m=128, steps=20, loss="softmax_nll", opt="adam", lr=1e-3, seed=0, data_seed=123),
    "A_sgd": dict(widths=[784, 256, 128, 10], m=128, steps=20, loss="softmax_nll", opt="sgd", lr=1e-2, seed=0, data_seed=123),
    "A_ragged": dict(widths=[784, 256, 128, 10], m=80, steps=3, loss="softmax_nll", opt="adam", lr=1e-3, seed=0, data_seed=321),
    "D_adam": dict(widths=[784, 256, 128, 10], m=1024, steps=5, loss="softmax_nll", opt="adam", lr=1e-3, seed=0, data_seed=456),
    "C_small": dict(widths=[256, 256, 256], m=64, steps=5, loss="mse", opt="adam", lr=1e-3, seed=0, data_seed=789),
    # the reference's OWN example net (examples/mnist/run.py:59-69): five Dense layers, hidden widths that are neither
    # multiples of 16 nor the 128 -> 10 head the benchmark shape ends in
    "R_example": dict(widths=[784, 200, 100, 70, 30, 10], m=128, steps=10, loss="softmax_nll", opt="adam", lr=1e-3, seed=0,
                      data_seed=135),
    # ... the same net at config D's global batch: the single-process answer of its data-parallel runs and of the row-block
    # step forms (256 / 512 / 1024 rows per GPU)
    "R_example_D": dict(widths=[784, 200, 100, 70, 30, 10], m=1024, steps=5, loss="softmax_nll", opt="adam", lr=1e-3, seed=0,
                        data_seed=246),
}


def sample_idx(n, k=512):
    """Fixed pseudo-random subset of a flat array (keeps fixtures small)."""
    return np.sort(np.random.RandomState(n % 100003).choice(n, size=min(k, n), replace=False))


def summarize(arr):
    flat = np.asarray(arr, dtype=np.float64).ravel()
    idx = sample_idx(flat.size)
    return {"sum": flat.sum(), "abs_sum": np.abs(flat).sum(), "l2": np.sqrt((flat ** 2).sum()),
            "idx": idx, "sample": flat[idx]}


# ---------------------------------------------------------------------------- reference runs
def run_reference(cfg, ref):
    rt, rops, rlayers, rlosses, ropt, rmodel, rnn, _ = ref
    w = cfg["widths"]
    np.random.seed(cfg["seed"])
    layers = []
    for i in range(len(w) - 1):
        layers.append(rlayers.Dense(w[i + 1], num_in=w[i]))        # eager init, layer order
        if i < len(w) - 2:
            layers.append(rlayers.ReLU())
    net = rnn.Net(layers)
    opt = ropt.Adam(lr=cfg["lr"]) if cfg["opt"] == "adam" else ropt.SGD(lr=cfg["lr"])
    loss_layer = rlosses.SoftmaxCrossEntropyLoss()
    model = rmodel.Model(net=net, loss=loss_layer, optimizer=opt)
    init = [{k: np.array(v.values) for k, v in l.params.items()} for l in layers if l.params]
    rec = {"loss": [], "logits": {}, "grads0": None, "argmax": {}}
    data = batches(cfg["data_seed"], cfg["steps"], cfg["m"], w[0], w[-1], cfg["loss"])
    for s, (x, y) in enumerate(data):
        model.zero_grad()
        pred = model.forward(rt.Tensor(x))
        if cfg["loss"] == "softmax_nll":
            loss = loss_layer.loss(pred, rt.Tensor(y))
        else:
            err = pred - rt.Tensor(y)
            loss = (err ** 2).sum() / cfg["m"]
        loss.backward()
        if s == 0:
            rec["grads0"] = [{k: np.array(v.grad) for k, v in l.params.items()} for l in layers if l.params]
        model.step()
        rec["loss"].append(float(loss.values))
        if s in (0, 1, cfg["steps"] - 1):
            rec["logits"][s] = np.array(pred.values)
            rec["argmax"][s] = np.argmax(np.array(pred.values), axis=1)
    final = [{k: np.array(v.values) for k, v in l.params.items()} for l in layers if l.params]
    return init, rec, final


def run_oracle(cfg):
    from oracle import ref_nn
    w = cfg["widths"]
    np.random.seed(cfg["seed"])
    layers = ref_nn.build_mlp(w)
    opt = ref_nn.Adam(lr=cfg["lr"]) if cfg["opt"] == "adam" else ref_nn.SGD(lr=cfg["lr"])
    loss_fn = ref_nn.softmax_nll if cfg["loss"] == "softmax_nll" else ref_nn.squared_error
    losses, logits = [], {}
    for s, (x, y) in enumerate(batches(cfg["data_seed"], cfg["steps"], cfg["m"], w[0], w[-1], cfg["loss"])):
        lv, pred = ref_nn.train_step(layers, opt, loss_fn, x, y)
        losses.append(float(lv))
        logits[s] = np.array(pred)
    final = [{k: np.array(v.values) for k, v in l.params.items()} for l in layers if l.params]
    return losses, logits, final


def run_closed_form(cfg, init):
    from oracle.closed_form import ClosedFormMLP
    w = cfg["widths"]
    mlp = ClosedFormMLP([p["w"] for p in init], [p["b"] for p in init], loss=cfg["loss"],
                        optimizer=cfg["opt"], lr=cfg["lr"])
    losses, grads0 = [], None
    for s, (x, y) in enumerate(batches(cfg["data_seed"], cfg["steps"], cfg["m"], w[0], w[-1], cfg["loss"])):
        loss, _, gW, gb = mlp.step(x, y)
        if s == 0:
            grads0 = (gW, gb)
        losses.append(loss)
    return losses, grads0, mlp


def trajectory_fixture(name, cfg, ref):
    init, rec, final = run_reference(cfg, ref)
    # --- pin the oracles on the reference
    o_losses, o_logits, o_final = run_oracle(cfg)
    assert np.array_equal(np.array(o_losses), np.array(rec["loss"])), "%s: ref_nn loss differs from the reference" % name
    for s, z in rec["logits"].items():
        assert np.array_equal(o_logits[s], z), "%s: ref_nn logits differ at step %d" % (name, s)
    for a, b in zip(o_final, final):
        for k in a:
            assert np.array_equal(a[k], b[k]), "%s: ref_nn final params differ" % name
    c_losses, (gW, gb), mlp = run_closed_form(cfg, init)
    # the reference's step-0 forward runs in float32 (params are float32 until the first `+=`),
    # the closed form is float64 throughout: agreement is to float32 round-off of that one forward
    np.testing.assert_allclose(c_losses, rec["loss"], rtol=1e-6, err_msg="%s closed-form loss" % name)
    for l, g in enumerate(rec["grads0"]):
        for mine, theirs in ((gW[l], g["w"]), (gb[l], g["b"])):
            scale = np.abs(theirs).max()
            assert np.abs(mine - theirs).max() <= 2e-5 * scale, "%s closed-form grad layer %d" % (name, l)
    for l, p in enumerate(final):
        np.testing.assert_allclose(mlp.W[l], p["w"], rtol=0, atol=1e-4)
    # --- write the fixture
    out = {"loss": np.array(rec["loss"], dtype=np.float64),
           "config": np.array(json.dumps(cfg))}
    for s, z in rec["logits"].items():
        if z.size <= 4096:
            out["logits_%d" % s] = z.astype(np.float64)
        else:
            summ = summarize(z)
            for k, v in summ.items():
                out["logits_%d_%s" % (s, k)] = v
        out["argmax_%d" % s] = rec["argmax"][s].astype(np.int64)
    for l, (p0, g0, p1) in enumerate(zip(init, rec["grads0"], final)):
        for k in ("w", "b"):
            out["init_%d%s_checksum" % (l, k)] = np.array([p0[k].astype(np.float64).sum(),
                                                             np.abs(p0[k].astype(np.float64)).sum()])
            for tag, arr in (("grad0", g0[k]), ("final", p1[k])):
                if arr.size <= 4096:
                    out["%s_%d%s" % (tag, l, k)] = np.asarray(arr, dtype=np.float64)
                else:
                    for kk, v in summarize(arr).items():
                        out["%s_%d%s_%s" % (tag, l, k, kk)] = v
    path = os.path.join(GOLDEN, "traj_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("  %-9s steps=%d loss[0]=%.6f loss[-1]=%.6f -> %s (%d KB)" % (
        name, cfg["steps"], rec["loss"][0], rec["loss"][-1], os.path.relpath(path, ROOT),
        os.path.getsize(path) // 1024))


# ---------------------------------------------------------------------------- op cases
class _RefOps(object):
    def __init__(self, rops):
        for n in ("exp", "log", "max", "maximum", "minimum", "reshape", "pad", "flatten", "clip", "sum"):
            setattr(self, n, getattr(rops, n))
        self.min = lambda obj, axis=None: rops.min_(rops.as_tensor(obj), axis=axis)


class OracleOps(object):
    """The same namespace over oracle/ref_autograd (also used by tests/test_oracle_golden.py)."""

    def __init__(self):
        from oracle import ref_autograd as ra
        self.exp, self.log, self.max, self.min = ra.exp, ra.log, ra.rmax, ra.rmin
        self.maximum, self.minimum = ra.maximum, ra.minimum
        self.reshape, self.pad, self.flatten, self.clip, self.sum = ra.reshape, ra.pad, ra.flatten, ra.clip, ra.rsum


def op_case_fixture(ref):
    import op_cases
    from oracle.ref_autograd import RefTensor
    rt, rops = ref[0], ref[1]
    out = {}
    for name, fn in op_cases.CASES.items():
        got = fn(rt.Tensor, _RefOps(rops))
        mine = fn(RefTensor, OracleOps())
        for k in got:
            a, b = np.asarray(got[k]), np.asarray(mine[k])
            assert a.shape == b.shape and np.array_equal(a, b), "oracle differs from reference: %s/%s" % (name, k)
        out[name] = {k: {"shape": list(np.asarray(v).shape), "data": np.asarray(v, dtype=np.float64).ravel().tolist()}
                     for k, v in got.items()}
    path = os.path.join(GOLDEN, "op_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("  %d op cases (reference == oracle, exact) -> %s" % (len(out), os.path.relpath(path, ROOT)))


# ---------------------------------------------------------------------------- eval path
def eval_fixture(ref):
    reval = ref[7]
    rs = np.random.RandomState(2024)
    logits = rs.randn(1000, 10).astype(np.float32)
    logits[::50, 3] = logits[::50].max(axis=1)          # forced ties: numpy argmax takes the FIRST maximum
    logits[::50, 7] = logits[::50, 3]
    targets = rs.randint(0, 10, 1000)
    pred = np.argmax(logits, axis=1)
    res = reval.AccEvaluator.evaluate(pred, targets)
    np.savez_compressed(os.path.join(GOLDEN, "eval.npz"), argmax=pred.astype(np.int64),
                        targets=targets.astype(np.int64), total_num=res["total_num"], hit_num=res["hit_num"],
                        accuracy=res["accuracy"])
    print("  eval: %s" % res)


OPTIMIZERS = {                   # name -> (reference class name, constructor kwargs)
    "momentum": ("Momentum", dict(lr=0.05, momentum=0.9)),
    "rmsprop": ("RMSProp", dict(lr=0.01, decay=0.99, momentum=0.0)),
    "rmsprop_mom": ("RMSProp", dict(lr=0.01, decay=0.9, momentum=0.5)),
    "adagrad": ("Adagrad", dict(lr=0.1)),
    "adadelta": ("Adadelta", dict(lr=1.0, decay=0.9)),
}


def optimizer_fixture(ref):
    """The four optimizers SURVEY §8(f)-4 names, straight from the reference's `_compute_step` (core/optimizer.py:
    82-164): six consecutive steps on a fixed gradient sequence (float64, like the reference's flat vector)."""
    ropt = ref[4]
    rs = np.random.RandomState(77)
    grads = rs.randn(6, 257) * np.array([1.0, 0.1, 3.0, 1e-3, 1.0, 0.5])[:, None]
    grads[2, ::17] = 0.0                                  # exact zeros: sqrt(eps) denominators
    out = {"grads": grads}
    for name, (cls, kw) in OPTIMIZERS.items():
        opt = getattr(ropt, cls)(**kw)
        out[name] = np.stack([np.asarray(opt._compute_step(g.copy()), dtype=np.float64) for g in grads])
    np.savez_compressed(os.path.join(GOLDEN, "optim_steps.npz"), **out)
    print("  optimizers: %s" % ", ".join(OPTIMIZERS))


INIT_SHAPES = [(30, 20), (1, 20), (6, 3, 3, 4)]
INITIALIZERS = {                 # name -> (reference class name, constructor kwargs)
    "normal": ("NormalInit", dict(mean=0.5, std=2.0)),
    "truncated_normal": ("TruncatedNormalInit", dict(mean=0.0, std=1.0)),
    "uniform": ("UniformInit", dict(a=-1.0, b=3.0)),
    "constant": ("ConstantInit", dict(val=3.1)),
    "zeros": ("ZerosInit", dict()),
    "xavier_uniform": ("XavierUniformInit", dict()),
    "xavier_normal": ("XavierNormalInit", dict()),
    "he_uniform": ("HeUniformInit", dict()),
    "he_normal": ("HeNormalInit", dict()),
}


def host_side_fixture():
    """SURVEY §8 a24 / a25, the host-side callers either side of the path: every initializer of core/initializer.py
    (`Initializer.__call__`, :17-19, on the global numpy RNG seeded with 123 before each draw) on three shapes, `get_fans`
    (:7-14), and the index order utils/data_iterator.py:22-34 produces for two epochs of 103 rows in batches of 32
    (shuffled, seed 7) and unshuffled."""
    import core.initializer as rinit
    from utils.data_iterator import BatchIterator
    out = {}
    for name, (cls, kw) in INITIALIZERS.items():
        for si, shape in enumerate(INIT_SHAPES):
            np.random.seed(123)
            t = getattr(rinit, cls)(**kw)(shape)
            out["%s_%d" % (name, si)] = np.asarray(t.values)
            out["%s_%d_dtype" % (name, si)] = np.array(str(np.asarray(t.values).dtype))
    out["fans"] = np.array([rinit.get_fans(s) for s in INIT_SHAPES + [(100, 10), (64, 5, 5, 128)]], dtype=np.int64)
    x = np.arange(103 * 2, dtype=np.float64).reshape(103, 2)
    y = np.arange(103, dtype=np.int64)
    for tag, shuffle in (("shuffled", True), ("ordered", False)):
        np.random.seed(7)
        it = BatchIterator(batch_size=32, shuffle=shuffle)
        order = [np.asarray(b.targets).copy() for _ in range(2) for b in it(x, y)]
        out["iter_%s_sizes" % tag] = np.array([len(o) for o in order], dtype=np.int64)
        out["iter_%s_targets" % tag] = np.concatenate(order)
    np.savez_compressed(os.path.join(GOLDEN, "host_side.npz"), **out)
    print("  initializers: %s; batch iterator orders" % ", ".join(INITIALIZERS))


def layer_fixture(ref):
    """Activation layers as objects: the reference's own `Tanh` — (1 - e^-x)/(1 + e^-x) = tanh(x/2),
    core/layers.py:83-89 — and `ReLU` (:92-98) run forward and backward on fixed inputs, float64 and float32; plus
    three SGD steps of a Dense-Tanh-Dense net under the softmax loss.  `Sigmoid` is absent: the reference's raises
    (SURVEY F7).  The oracle's restatement (ref_nn.Tanh) must agree bit for bit."""
    import synth
    from oracle import ref_nn
    from oracle.ref_autograd import RefTensor
    rt, rlayers, rlosses, ropt, rmodel, rnn = ref[0], ref[2], ref[3], ref[4], ref[5], ref[6]
    x, g = synth.layer_inputs()
    out = {}
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        for name, cls, mine in (("tanh", rlayers.Tanh, ref_nn.Tanh), ("relu", rlayers.ReLU, ref_nn.ReLU)):
            t = rt.Tensor(x.astype(dt), requires_grad=True)
            y = cls().forward(t)
            y.backward(g)
            o = RefTensor(x.astype(dt), requires_grad=True)
            yo = mine().forward(o)
            yo.backward(g)
            assert np.array_equal(np.asarray(y.values), yo.values) and np.array_equal(t.grad, o.grad), name
            out["%s_%s_out" % (name, tag)] = np.asarray(y.values, dtype=np.float64)
            out["%s_%s_grad" % (name, tag)] = np.asarray(t.grad, dtype=np.float64)
    # Dense(6) - Tanh - Dense(4), softmax loss, SGD lr 0.1, three steps on one batch
    rs = np.random.RandomState(17)
    bx = rs.randn(12, 9).astype(np.float32)
    by = np.eye(4)[rs.randint(0, 4, 12)]
    np.random.seed(11)
    net = rnn.Net([rlayers.Dense(6, num_in=9), rlayers.Tanh(), rlayers.Dense(4, num_in=6)])
    loss_layer = rlosses.SoftmaxCrossEntropyLoss()
    model = rmodel.Model(net=net, loss=loss_layer, optimizer=ropt.SGD(lr=0.1))
    np.random.seed(11)
    mine = [ref_nn.Dense(9, 6), ref_nn.Tanh(), ref_nn.Dense(6, 4)]
    mopt = ref_nn.SGD(lr=0.1)
    losses = []
    for s in range(3):
        model.zero_grad()
        loss = loss_layer.loss(model.forward(rt.Tensor(bx)), rt.Tensor(by))
        loss.backward()
        if s == 0:
            out["net_grad0_w0"] = np.array(net.layers[0].params["w"].grad)
        model.step()
        losses.append(float(loss.values))
        lv, _ = ref_nn.train_step(mine, mopt, ref_nn.softmax_nll, bx, by)
        assert float(lv) == losses[-1], "ref_nn Dense-Tanh-Dense step %d differs from the reference" % s
    out["net_loss"] = np.array(losses)
    out["net_final_w0"] = np.array(net.layers[0].params["w"].values, dtype=np.float64)
    out["net_final_w1"] = np.array(net.layers[2].params["w"].values, dtype=np.float64)
    assert np.array_equal(mine[0].params["w"].values, net.layers[0].params["w"].values)
    np.savez_compressed(os.path.join(GOLDEN, "layers.npz"), **out)
    print("  layers: Tanh / ReLU forward + vjp (f64, f32), Dense-Tanh-Dense 3 SGD steps, loss %.6f -> %.6f"
          % (losses[0], losses[-1]))


def epoch_fixture(ref):
    """SURVEY §8 a25 end to end: the reference's own loop (examples/mnist/run.py:45-93) — random_seed, Tensors of the
    whole dataset, a Net of lazily initialised Dense layers, BatchIterator (utils/data_iterator.py:22-34: per-epoch
    shuffle on the global RNG, ragged last batch), zero_grad / forward / loss / backward / step per batch, then per
    epoch `np.argmax(test_pred, axis=1)` on the Tensor and AccEvaluator — on 1000 synthetic rows (tests/synth.py),
    bs 128 -> 7 full batches + 104 rows, 2 epochs.  Only the MNIST download and the net's widths differ from run.py
    (784-256-128-10 = BASELINE.json's net instead of 200-100-70-30)."""
    import synth
    from oracle import ref_nn
    rt, rlayers, rlosses, ropt, rmodel, rnn, reval = ref[0], ref[2], ref[3], ref[4], ref[5], ref[6], ref[7]
    from utils.data_iterator import BatchIterator
    from utils.seeder import random_seed
    cfg = synth.EPOCH_CFG
    train_x, train_lab, pool_x, pool_lab = synth.epoch_dataset(cfg)
    train_y = np.eye(10)[np.array(train_lab).reshape(-1)]            # get_one_hot, run.py:27-28
    w = cfg["widths"]

    def run(test_x, test_lab, probe=None):
        """The loop of run.py:45-93.  probe: extra rows forwarded after each epoch (no RNG use, no state change) whose
        top-2 logit gaps are returned — how the well-conditioned evaluation rows are chosen."""
        random_seed(cfg["seed"])
        tx, ty, sx = rt.Tensor(train_x), rt.Tensor(train_y), rt.Tensor(test_x)
        layers = []
        for i in range(1, len(w)):
            layers.append(rlayers.Dense(w[i]))                        # lazy: shapes from the first batch
            if i < len(w) - 1:
                layers.append(rlayers.ReLU())
        net = rnn.Net(layers)
        model = rmodel.Model(net=net, loss=rlosses.SoftmaxCrossEntropyLoss(), optimizer=ropt.Adam(lr=cfg["lr"]))
        loss_layer = rlosses.SoftmaxCrossEntropyLoss()
        iterator = BatchIterator(batch_size=cfg["batch_size"])
        evaluator = reval.AccEvaluator()
        loss_list, sizes, preds, results, margins, probe_gaps = [], [], [], [], [], []
        for epoch in range(cfg["num_ep"]):
            for batch in iterator(tx, ty):
                model.zero_grad()
                pred = model.forward(batch.inputs)
                loss = loss_layer.loss(pred, batch.targets)
                loss.backward()
                model.step()
                loss_list.append(float(loss.values))
                sizes.append(len(batch.inputs))
            model.set_phase("TEST")
            test_pred = model.forward(sx)
            idx = np.argmax(test_pred, axis=1)                         # on the Tensor, as run.py:89 does
            assert np.array_equal(idx, np.argmax(test_pred.values, axis=1))
            results.append(evaluator.evaluate(idx, test_lab))
            preds.append(np.asarray(idx, dtype=np.int64))
            top2 = np.sort(np.asarray(test_pred.values), axis=1)[:, -2:]
            margins.append(float((top2[:, 1] - top2[:, 0]).min()))
            if probe is not None:
                z = np.asarray(model.forward(rt.Tensor(probe)).values)
                t2 = np.sort(z, axis=1)[:, -2:]
                probe_gaps.append(t2[:, 1] - t2[:, 0])
            model.set_phase("TRAIN")
        return loss_list, sizes, preds, results, margins, probe_gaps

    # pass 1: train, look at the whole candidate pool after each epoch; keep the first n_test rows whose top-2 gap
    # exceeds min_margin after BOTH epochs
    first = run(pool_x[:8], pool_lab[:8], probe=pool_x)
    gap = np.minimum.reduce(first[5])
    rows = np.nonzero(gap > cfg["min_margin"])[0][:cfg["n_test"]]
    assert len(rows) == cfg["n_test"], "only %d pool rows clear the margin" % len(rows)
    test_x, test_lab = pool_x[rows], pool_lab[rows]
    # pass 2: the run proper, evaluating those rows exactly like run.py does
    loss_list, sizes, preds, results, margins, _ = run(test_x, test_lab)
    assert loss_list == first[0], "the probe forward changed the training trajectory"
    assert min(margins) > cfg["min_margin"]
    # pass 3: the same run evaluating the WHOLE candidate pool, near-ties included — what a float64 device run must
    # reproduce row for row (tests: epoch_loop_float64_all_pool_rows)
    p_losses, _, p_preds, p_results, p_margins, _ = run(pool_x, pool_lab)
    assert p_losses == loss_list
    # --- pin the oracle's restatement of the loop
    random_seed(cfg["seed"])
    o_losses, o_preds, o_results = ref_nn.train_epochs(w, train_x, train_y, test_x, test_lab, cfg["num_ep"],
                                                       cfg["batch_size"], cfg["lr"])
    assert o_losses == loss_list, "ref_nn.train_epochs losses differ from the reference loop"
    assert all(np.array_equal(a, b) for a, b in zip(o_preds, preds)) and o_results == results
    np.savez_compressed(os.path.join(GOLDEN, "epoch.npz"), loss=np.array(loss_list), batch_sizes=np.array(sizes),
                        test_rows=rows.astype(np.int64),
                        argmax=np.stack(preds), hit_num=np.array([r["hit_num"] for r in results], dtype=np.int64),
                        total_num=np.array([r["total_num"] for r in results], dtype=np.int64),
                        accuracy=np.array([r["accuracy"] for r in results]), min_top2_margin=np.array(margins),
                        pool_argmax=np.stack(p_preds), pool_hit_num=np.array([r["hit_num"] for r in p_results], dtype=np.int64),
                        pool_min_top2_margin=np.array(p_margins),
                        config=np.array(json.dumps(cfg)))
    print("  epoch (whole pool, %d rows): eval %s, min top-2 margin %.2e" % (len(pool_x), p_results, min(p_margins)))
    print("  epoch: %d steps (batch sizes %s), loss %.4f -> %.4f, eval %s on %d of %d pool rows, min top-2 margin %.2e"
          % (len(loss_list), sorted(set(sizes)), loss_list[0], loss_list[-1], results, len(rows), len(pool_x), min(margins)))


def reference_own_tests():
    """The reference's own unit tests must pass in this container (pins the import itself)."""
    import subprocess
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=REF)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider",
                        os.path.join(REF, "test", "test_autograd.py")], capture_output=True, text=True, env=env,
                       cwd="/tmp")
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]
    print("  reference test/test_autograd.py: %s" % tail)
    assert r.returncode == 0, r.stdout + r.stderr


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ref = import_reference()
    print("generating golden fixtures from %s (numpy %s)" % (REF, np.__version__))
    only = sys.argv[1:]
    if only == ["optim"]:                                   # just the optimizer-step fixture
        optimizer_fixture(ref)
        return
    if only == ["host"]:                                    # just the initializer / iterator fixture
        host_side_fixture()
        return
    if only == ["layers"]:
        layer_fixture(ref)
        return
    if only == ["epoch"]:
        epoch_fixture(ref)
        return
    reference_own_tests()
    op_case_fixture(ref)
    optimizer_fixture(ref)
    host_side_fixture()
    layer_fixture(ref)
    epoch_fixture(ref)
    for name, cfg in CONFIGS.items():
        if only and name not in only:
            continue
        trajectory_fixture(name, cfg, ref)
    eval_fixture(ref)
    print("done")


if __name__ == "__main__":
    main()
