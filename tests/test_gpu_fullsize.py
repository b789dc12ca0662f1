"""Full-size checks on the MI355X (BASELINE.json configs[2]: Dense 4096->4096->4096, bs 512): the whole-step
trainer against the float64 closed-form oracle computed on the host, plus size-independent GEMM properties at
the sizes the roofline is quoted on."""

import numpy as np
import pytest

import tinynn_autograd_amd as tn
from oracle.closed_form import ClosedFormMLP
from tinynn_autograd_amd.fused import MLPTrainer


@pytest.mark.gpu
def test_config_C_two_steps_match_float64_closed_form():
    rs = np.random.RandomState(4)
    widths, m = [4096, 4096, 4096], 512
    a = np.sqrt(6.0 / (4096 + 4096))
    W = [rs.uniform(-a, a, (4096, 4096)).astype(np.float32) for _ in range(2)]
    B = [np.zeros((1, 4096), np.float32) for _ in range(2)]
    x = rs.rand(m, 4096).astype(np.float32)
    trainer = MLPTrainer(widths, m, loss="mse", optimizer="adam", lr=1e-3)
    trainer.set_parameters([{"w": W[i], "b": B[i]} for i in range(2)])
    oracle = ClosedFormMLP(W, B, loss="mse", optimizer="adam", lr=1e-3)
    xd = tn.asarray(x)
    # ---- step 0: loss, ReLU mask, gradients
    loss = float(trainer.step(xd, xd))
    dev_mask = ~np.signbit(np.asarray(trainer.activation(0, m)))          # the mask the device used (sign bit of a)
    ref_loss, _, gW, gb = oracle.loss_and_grads(x, x, masks=[dev_mask])
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-5)
    z0 = oracle.last_pre_activations[0]
    flipped = dev_mask != (z0 >= 0)
    # ReLU' is discontinuous: a float32 pre-activation may land on the other side of 0 only where the float64
    # value is within float32 round-off of it (K = 4096 products of magnitude <= 0.04)
    assert flipped.sum() <= 64 and (np.abs(z0[flipped]) < 1e-5).all(), (flipped.sum(), np.abs(z0[flipped]).max())
    for l in range(2):
        g = np.asarray(trainer.grad_view(l, "w"))
        assert np.abs(g - gW[l]).max() <= 1e-5 * np.abs(gW[l]).max(), "dW%d" % l
        gbd = np.asarray(trainer.grad_view(l, "b"))
        assert np.abs(gbd - gb[l]).max() <= 1e-5 * np.abs(gb[l]).max(), "db%d" % l
    # ---- step 1 continues from the device's own (float32, Adam-updated) parameters: Adam's m/(sqrt(v)+eps) is
    # sign-like where |g| ~ 0, so 33.5 M parameters each moved by up to lr in a rounding-dependent direction;
    # the loss still tracks the float64 trajectory to 1e-4
    oracle.step(x, x)
    loss1 = float(trainer.step(xd, xd))
    ref_loss1, _, _, _ = oracle.loss_and_grads(x, x)
    np.testing.assert_allclose(loss1, ref_loss1, rtol=1e-4)
    for l in range(2):
        p = np.asarray(trainer.param_view(l, "w"))
        assert np.abs(p - W[l]).max() > 1e-4                 # the parameters really moved
        assert np.abs(p - oracle.W[l]).max() <= 2.5e-3       # and stay within ~2 Adam steps (lr = 1e-3) of float64
    # ---- the same two steps with Adam in the epilogues of the dW GEMMs and the weight gradients never stored (what
    # bench.py times for this config): losses, parameters and both moments bit-identical to the run above
    ref = [np.asarray(t).copy() for t in (trainer.params, trainer.adam_m, trainer.adam_v)]
    del trainer
    t2 = MLPTrainer(widths, m, loss="mse", optimizer="adam", lr=1e-3).keep_grads(False)
    t2.set_parameters([{"w": W[i], "b": B[i]} for i in range(2)])
    assert float(t2.step(xd, xd)) == loss and float(t2.step(xd, xd)) == loss1
    for got, want in zip((t2.params, t2.adam_m, t2.adam_v), ref):
        assert np.array_equal(np.asarray(got), want)


@pytest.mark.gpu
def test_dw_gemm_with_adam_epilogue_equals_gemm_then_adam():
    """tnn_gemm_tn_adam against tnn_gemm_tn_colsum + tnn_adam: p / m / v bit-identical over two steps on a shape the tiled
    kernel takes (interior and ragged tiles), on MNIST-size and odd shapes (its fallback), with and without the stored
    gradient."""
    from tinynn_autograd_amd import _lib
    lib = _lib.get()
    rs = np.random.RandomState(6)
    for (M, N, K) in ((4096, 4096, 512), (2000, 1100, 96), (784, 256, 128), (33, 20, 7)):
        a = tn.asarray(rs.uniform(-1, 1, (K, M)).astype(np.float32))
        gmat = tn.asarray((rs.uniform(-1, 1, (K, N)) * 1e-2).astype(np.float32))
        p0 = rs.randn(M, N).astype(np.float32)
        res = []
        for fused, keep in ((False, True), (True, True), (True, False)):
            P, M_, V_ = tn.asarray(p0), tn.zeros((M, N)), tn.zeros((M, N))
            G = tn.asarray(np.full((M, N), 7.0, np.float32))
            pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
            for _ in range(2):
                lib.adam_tick(pows._ptr, 0.9, 0.999)
                if fused:
                    lib.gemm_tn_adam(M, N, K, a._ptr, M, gmat._ptr, N, G._ptr if keep else None, P._ptr, M_._ptr, V_._ptr,
                                     1e-3, 0.9, 0.999, 1e-8, pows._ptr, _lib.F32)
                else:
                    lib.gemm_tn_colsum(M, N, K, a._ptr, M, gmat._ptr, N, G._ptr, N, None, _lib.F32)
                    lib.adam_ex(P._ptr, G._ptr, M_._ptr, V_._ptr, M * N, 1e-3, 0.9, 0.999, 1e-8, pows._ptr, None,
                                _lib.F32, 0, None, None)
            res.append([np.asarray(t) for t in (P, M_, V_, pows, G)])
        for got in res[1:]:
            for name, x0, x1 in zip(("p", "m", "v", "pows"), res[0], got):
                assert np.array_equal(x0, x1), (name, M, N, K)
        assert np.array_equal(res[0][4], res[1][4]) and (res[2][4] == 7.0).all()
        assert np.abs(res[0][0] - p0).max() > 5e-4


@pytest.mark.gpu
def test_dw_gemm_with_adam_and_bias_in_one_launch():
    """tnn_gemm_tn_adam_bias (dW + Adam on W + db + Adam on b from ONE launch: the workgroups of tile row 0 sum the columns of
    dz from the fragments they stream) against the launches it replaces — tnn_gemm_tn_adam + tnn_reduce + tnn_adam_ex: weights,
    bias, both moments of each and db bit-identical over two steps (the column sums accumulate in float64 on both sides);
    tiled shapes (interior + ragged tiles) and the fallback shapes.  tnn_mse_fwd_bwd_tick: loss to both destinations,
    gradient, and the beta powers advanced once."""
    from tinynn_autograd_amd import _lib
    lib = _lib.get()
    rs = np.random.RandomState(16)
    for (M, N, K) in ((4096, 4096, 512), (2000, 1100, 96), (784, 256, 128)):
        a = tn.asarray(rs.uniform(-1, 1, (K, M)).astype(np.float32))
        gmat = tn.asarray((rs.uniform(-1, 1, (K, N)) * 1e-2).astype(np.float32))
        p0, pb0 = rs.randn(M, N).astype(np.float32), rs.randn(N).astype(np.float32)
        res = []
        for fused in (False, True):
            P, M_, V_ = tn.asarray(p0), tn.zeros((M, N)), tn.zeros((M, N))
            PB, MB, VB, DB = tn.asarray(pb0), tn.zeros((N,)), tn.zeros((N,)), tn.zeros((N,))
            pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
            for _ in range(2):
                lib.adam_tick(pows._ptr, 0.9, 0.999)
                if fused:
                    lib.gemm_tn_adam_bias(M, N, K, a._ptr, M, gmat._ptr, N, None, P._ptr, M_._ptr, V_._ptr, DB._ptr, PB._ptr,
                                          MB._ptr, VB._ptr, 1e-3, 0.9, 0.999, 1e-8, pows._ptr, _lib.F32)
                else:
                    lib.gemm_tn_adam(M, N, K, a._ptr, M, gmat._ptr, N, None, P._ptr, M_._ptr, V_._ptr, 1e-3, 0.9, 0.999, 1e-8,
                                     pows._ptr, _lib.F32)
                    lib.reduce(_lib.RSUM, gmat._ptr, DB._ptr, 1, K, N, _lib.F32)
                    lib.adam_ex(PB._ptr, DB._ptr, MB._ptr, VB._ptr, N, 1e-3, 0.9, 0.999, 1e-8, pows._ptr, None, _lib.F32, 0,
                                None, None)
            res.append([np.asarray(t) for t in (P, M_, V_, PB, MB, VB, DB)])
        tiled = M * N >= 2000 * 1100        # the fallback's latency kernel sums dz's columns in float32: equal to round-off
        for name, x0, x1 in zip(("p", "m", "v", "pb", "mb", "vb", "db"), res[0], res[1]):
            if tiled or name in ("p", "m", "v"):
                assert np.array_equal(x0, x1), (name, M, N, K, np.abs(x0 - x1).max())
            else:
                np.testing.assert_allclose(x1, x0, rtol=2e-6, atol=2e-6 * np.abs(x0).max(), err_msg="%s %s" % (name, (M, N, K)))
        ref_db = np.asarray(gmat, dtype=np.float64).sum(0)
        assert np.abs(res[1][6] - ref_db).max() <= 1e-6 * np.abs(ref_db).max()
        assert np.abs(res[1][3] - pb0).max() > 5e-4
    n, mg = 512 * 4096, 512
    pr, yy = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    PR, YY, DP = tn.asarray(pr), tn.asarray(yy), tn.empty((n,))
    l1, l2 = tn.empty(()), tn.empty(())
    pows = tn.asarray(np.array([0.5, 0.25, 0, 0]), dtype=np.float64)
    lib.mse_fwd_bwd_tick(PR._ptr, YY._ptr, n, mg, l1._ptr, l2._ptr, DP._ptr, _lib.F32, pows._ptr, 0.9, 0.999)
    e = pr.astype(np.float64) - yy
    np.testing.assert_allclose(float(l1), (e ** 2).sum() / mg, rtol=1e-6)
    assert float(l1) == float(l2)
    np.testing.assert_allclose(np.asarray(DP), (2.0 * e / mg).astype(np.float32), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.asarray(pows)[:2], [0.45, 0.24975], rtol=1e-15)


@pytest.mark.gpu
def test_gemm_4096_against_float64_and_linearity():
    rs = np.random.RandomState(5)
    M, N, K = 512, 4096, 4096
    a = rs.uniform(-1, 1, (M, K)).astype(np.float32)
    b = rs.uniform(-1, 1, (K, N)).astype(np.float32)
    A, B = tn.asarray(a), tn.asarray(b)
    c = np.asarray(A @ B, dtype=np.float64)
    rows = rs.choice(M, 24, replace=False)
    ref = a[rows].astype(np.float64) @ b.astype(np.float64)
    bound = np.abs(a[rows]).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(c[rows] - ref) <= 1e-6 * bound).all()              # f32 round-off class (guide: ~3.5e-7 at K=4096)
    # NT and TN on the same data: (A B)^T == B^T A^T
    ct = np.asarray(tn.asarray(np.ascontiguousarray(b.T)) @ tn.asarray(np.ascontiguousarray(a.T)), dtype=np.float64)
    assert (np.abs(ct.T - c) <= 2e-6 * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64))).all()
    nt = np.asarray(A @ tn.asarray(np.ascontiguousarray(b.T)).T, dtype=np.float64)
    assert np.array_equal(nt, c) or np.abs(nt - c).max() <= 1e-3     # same products, possibly another summation order
    # linearity in A
    a2 = rs.uniform(-1, 1, (M, K)).astype(np.float32)
    lhs = np.asarray((A + tn.asarray(a2)) @ B, dtype=np.float64)
    rhs = c + np.asarray(tn.asarray(a2) @ B, dtype=np.float64)
    assert np.abs(lhs - rhs).max() <= 1e-2                            # |values| ~ 40, K = 4096


@pytest.mark.gpu
def test_fused_adam_33M_matches_numpy():
    from tinynn_autograd_amd import _lib
    n = 33562624
    rs = np.random.RandomState(6)
    p0 = rs.randn(n).astype(np.float32); g = (rs.randn(n) * 1e-2).astype(np.float32)
    P, G = tn.asarray(p0), tn.asarray(g)
    Mo, Vo = tn.zeros((n,)), tn.zeros((n,))
    pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    m = np.zeros(n); v = np.zeros(n); p = p0.astype(np.float64)
    for t in (1, 2):
        _lib.get().adam(P._ptr, G._ptr, Mo._ptr, Vo._ptr, n, lr, b1, b2, eps, pows._ptr, None, _lib.F32)
        m += (1 - b1) * (g - m); v += (1 - b2) * (g.astype(np.float64) ** 2 - v)
        p += -lr * (m / (1 - b1 ** t)) / (np.sqrt(v / (1 - b2 ** t)) + eps)
    np.testing.assert_allclose(np.asarray(P), p, rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.asarray(pows)[:2], [b1 ** 2, b2 ** 2], rtol=1e-14)


@pytest.mark.gpu
def test_gemm_fuzz_against_numpy():
    """tools/gemm_fuzz.py: random shapes / layouts / epilogues through every fp32 GEMM entry point, the fused Dense
    backward and the first-layer backward with Adam folded in, against float64 numpy."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_fuzz.py"), "120", "3"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "cases ok" in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
def test_example_learns_the_teacher_labels():
    """End to end: examples/mnist_run.py --trainer (resident dataset, per-epoch device shuffle, one hipGraph per epoch,
    device argmax + AccEvaluator) on the synthetic linear-teacher data must LEARN — accuracy far above the 10 % of
    chance after two epochs."""
    import os
    import re
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tinynn-autograd_amd", "examples", "mnist_run.py"), "--num_ep", "2",
                        "--trainer", "--seed", "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    acc = [float(a) for a in re.findall(r"'accuracy': ([0-9.]+)", r.stdout)]
    assert len(acc) == 2 and acc[-1] > 0.7, r.stdout[-2000:]


@pytest.mark.gpu
def test_epoch_loop_at_full_size_properties():
    """The reference's loop (examples/mnist/run.py:76-93, utils/data_iterator.py:22-34) at BASELINE's size — 50,000 rows,
    bs 128: 390 full batches + the ragged 80-row batch, 2 epochs — through size-independent properties: the epoch buffer
    is a PERMUTATION of the dataset (row checksums as a multiset) drawn from numpy's global RNG exactly like the reference's
    shuffle; a seeded run is deterministic bit for bit (two runs, the replayed epoch graph included); the trainer path,
    the recorded op-level loop and the eager op-level loop tell the same story (first-epoch losses within 1e-4 over the first 30
    float32 Adam steps and within 1e-2 over all 391, identical step counts, accuracies within a point); the loss falls and the accuracy rises."""
    from tinynn_autograd_amd.core.tensor import Tensor
    from tinynn_autograd_amd.examples import mnist_run
    from tinynn_autograd_amd.utils.data_iterator import BatchIterator
    (train_x, train_y), (test_x, test_y), _ = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
    # ---- the gather is a permutation, and the one the reference would draw
    it = BatchIterator(batch_size=128, reuse_buffers=True)
    tx, ty = Tensor(train_x), Tensor(mnist_run.get_one_hot(train_y, 10))
    np.random.seed(5)
    want = np.arange(50000); np.random.shuffle(want)
    np.random.seed(5)
    batches = list(it(tx, ty))
    assert len(batches) == 391 and len(batches[-1].inputs) == 80
    got = np.concatenate([np.asarray(b.inputs.values) for b in batches])
    assert np.array_equal(got, train_x[want])
    words = lambda a: np.sort(np.ascontiguousarray(a).view(np.uint32).sum(axis=1, dtype=np.uint64))   # one integer checksum per row
    assert np.array_equal(words(got), words(train_x))
    del batches, it, tx, ty
    # ---- determinism and agreement of the three paths
    runs = {}
    for name, kw in (("trainer", {"trainer": True}), ("trainer_again", {"trainer": True}), ("capture", {"capture": True}), ("eager", {})):
        np.random.seed(0)
        runs[name] = mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 2, 128, 1e-3, **kw)
    losses, preds, results = runs["trainer"]
    assert len(losses) == 2 * 391 and np.isfinite(losses).all()
    assert losses == runs["trainer_again"][0] and all(np.array_equal(a, b) for a, b in zip(preds, runs["trainer_again"][1]))
    assert np.mean(losses[-50:]) < np.mean(losses[:50]) and results[1]["accuracy"] > results[0]["accuracy"] > 0.5
    for other in ("capture", "eager"):
        o_losses, _, o_results = runs[other]
        assert len(o_losses) == 2 * 391
        np.testing.assert_allclose(o_losses[:30], losses[:30], rtol=1e-4, err_msg=other)      # same arithmetic, other summation order:
        np.testing.assert_allclose(o_losses[:391], losses[:391], rtol=1e-2, err_msg=other)    # ... amplified by 391 float32 Adam steps
        for ep in range(2):
            assert o_results[ep]["total_num"] == 10000 and abs(o_results[ep]["accuracy"] - results[ep]["accuracy"]) < 0.01, (other, ep)
    assert runs["capture"][0][:391] == runs["eager"][0][:391]          # epoch 0 of the recorded path IS the eager path
