"""BASELINE.json configs[4] AT ITS STATED SIZE on the MI355X: 8192-wide 4-layer MLP, bf16 storage, fp32 accumulate,
fp32 master weights + Adam.  No reference file exists for this path (the reference is float64 numpy); the yardstick
is float64 numpy on the SAME bf16-rounded operands:

  * each GEMM form of the step (forward NT + bias + ReLU with the mask in the sign bit, dX NT + mask, dW with
    K = 512 into fp32) at 512 x 8192 x 8192 on sampled rows, error bound 2e-6 * (|A| |B|^T) (fp32 accumulation
    class) plus one bf16 rounding where the output is bf16;
  * one whole trainer step: loss, sampled rows of every dW, every db, against (a) a float64 host model that rounds to
    bf16 exactly where the device stores bf16 (activations, dz) — tight bars — and (b) oracle/closed_form.py
    (pure float64, no rounding) — the 2^-9-per-tensor bars of tests/test_gpu_bf16.py;
  * adam_master_bf16_2d at 8192 x 8192: bit-equal to the flat kernel, W^T copy == W.T.
"""

import os

import numpy as np
import pytest

import tinynn_autograd_amd as tn
from tinynn_autograd_amd import bf16

W8, M8 = 8192, 512


def _rand_bf16(rs, shape, lo=-1.0, hi=1.0):
    return bf16.round_to_bf16(rs.uniform(lo, hi, shape).astype(np.float32))


@pytest.mark.gpu
def test_config_E_gemm_forms_at_full_size():
    rs = np.random.RandomState(81)
    act = _rand_bf16(rs, (M8, W8), 0.0, 1.0)                      # a_{l-1}  [rows, in]
    w = _rand_bf16(rs, (W8, W8), -0.03, 0.03)                     # W_l      [in, out]
    bias = rs.randn(W8).astype(np.float32) * 0.1
    rows = np.sort(rs.choice(M8, 24, replace=False))
    A16, W16 = bf16.to_bf16(act), bf16.to_bf16(w)
    WT16 = bf16.transpose(W16)
    assert np.array_equal(np.asarray(WT16), np.asarray(W16).T)    # 8192 x 8192 bf16 transpose, bit-exact
    a64, w64 = act[rows].astype(np.float64), w.astype(np.float64)

    # ---- forward: z = a W + b, ReLU, mask in the sign bit of zero, bf16 out (A = a [rows,in], B = W^T [out,in])
    y16 = bf16.gemm_nt(A16, WT16, out_dtype=np.uint16, bias=tn.asarray(bias), relu=True, relu_sign=True)
    z = a64 @ w64 + bias
    bound = np.abs(a64) @ np.abs(w64) + np.abs(bias)
    raw = np.asarray(y16)[rows]
    y = np.asarray(bf16.to_f32(y16))[rows].astype(np.float64)
    assert (np.abs(y - np.maximum(z, 0)) <= 4e-3 * np.abs(z) + 2e-6 * bound).all(), np.abs(y - np.maximum(z, 0)).max()
    clearly_neg, clearly_pos = z < -2e-6 * bound, z > 2e-6 * bound
    assert ((raw[clearly_neg] & 0x8000) != 0).all() and ((raw[clearly_pos] & 0x8000) == 0).all()
    assert clearly_neg.sum() > 1000 and clearly_pos.sum() > 1000  # both branches really exercised

    # ---- plain fp32 output of the same product (fp32 accumulation over K = 8192)
    c = np.asarray(bf16.gemm_nt(A16, WT16), dtype=np.float64)[rows]
    assert (np.abs(c - a64 @ w64) <= 2e-6 * (np.abs(a64) @ np.abs(w64))).all()

    # ---- dX: dz_{l-1} = (dz_l W_l^T) * mask, bf16 out (A = dz [rows,out], B = W [in,out])
    dz = _rand_bf16(rs, (M8, W8), -1e-2, 1e-2)
    D16 = bf16.to_bf16(dz)
    dx16 = bf16.gemm_nt(D16, W16, out_dtype=np.uint16, mask=y16)
    d64 = dz[rows].astype(np.float64)
    ref = np.where((raw & 0x8000) != 0, 0.0, d64 @ w64.T)
    got = np.asarray(bf16.to_f32(dx16))[rows].astype(np.float64)
    assert (np.abs(got - ref) <= 4e-3 * np.abs(ref) + 2e-6 * (np.abs(d64) @ np.abs(w64).T)).all()
    assert ((got == 0) == ((raw & 0x8000) != 0) | (ref == 0)).all()

    # ---- dW = a^T dz: M = N = 8192, K = 512, fp32 out (A = a^T [in,rows], B = dz^T [out,rows])
    AT16, DT16 = bf16.transpose(A16), bf16.transpose(D16)
    dw = np.asarray(bf16.gemm_nt(AT16, DT16), dtype=np.float64)
    in_rows = np.sort(rs.choice(W8, 24, replace=False))
    aT, d_all = act[:, in_rows].T.astype(np.float64), dz.astype(np.float64)
    assert (np.abs(dw[in_rows] - aT @ d_all) <= 2e-6 * (np.abs(aT) @ np.abs(d_all)) + 1e-30).all()
    # a size-independent property on the whole 8192 x 8192 result: column sums of dW = (sum_rows a)^T dz
    colsum_ref = act.astype(np.float64).sum(1) @ d_all                              # sum over `in` of dW[in, :]
    colsum_bound = np.abs(act).astype(np.float64).sum(1) @ np.abs(d_all)
    assert (np.abs(dw.sum(0) - colsum_ref) <= 2e-6 * colsum_bound).all()


@pytest.mark.gpu
def test_config_E_adam_2d_at_full_size_equals_flat_kernel():
    from tinynn_autograd_amd import _lib
    rs = np.random.RandomState(82)
    R = C = W8
    p0 = rs.uniform(-0.03, 0.03, (R, C)).astype(np.float32)
    g = (rs.randn(R, C) * 1e-3).astype(np.float32)
    g[::97, ::89] = 0.0                                                             # exact zeros: eps-only denominators
    out = []
    for tiled in (False, True):
        P, G, M_, V_ = tn.asarray(p0), tn.asarray(g), tn.zeros((R, C)), tn.zeros((R, C))
        W16, WT16 = tn.empty((R, C), np.uint16), tn.empty((C, R), np.uint16)
        pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
        for _ in range(2):
            if tiled:
                _lib.get().adam_master_bf16_2d(P._ptr, G._ptr, M_._ptr, V_._ptr, W16._ptr, WT16._ptr, R, C,
                                               1e-3, 0.9, 0.999, 1e-8, pows._ptr, 1)
            else:
                _lib.get().adam_master_bf16(P._ptr, G._ptr, M_._ptr, V_._ptr, W16._ptr, R * C,
                                            1e-3, 0.9, 0.999, 1e-8, pows._ptr)
        res = [np.asarray(t) for t in (P, M_, V_, W16)] + [np.asarray(pows)]
        if tiled:
            assert np.array_equal(np.asarray(WT16), res[3].T)                       # the W^T working copy
            assert np.array_equal(np.asarray(bf16.to_f32(W16)), bf16.round_to_bf16(res[0]))
        out.append(res)
        del P, G, M_, V_, W16, WT16
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    # and against float64 numpy after the two steps (same gradient twice)
    g64 = g.astype(np.float64)
    m = v = 0.0
    p = p0.astype(np.float64)
    for t in (1, 2):
        m = m + 0.1 * (g64 - m)
        v = v + 0.001 * (g64 ** 2 - v)
        p = p - 1e-3 * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-8)
    np.testing.assert_allclose(out[1][0], p, rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_config_E_trainer_step_at_full_size():
    """One full step of the 8192-8192-8192-8192-8192 bf16 trainer (268 M parameters, bs 512, sum-of-squares loss, Adam)."""
    from oracle.closed_form import ClosedFormMLP
    from tinynn_autograd_amd.fused import MLPTrainer
    rs = np.random.RandomState(83)
    L, widths = 4, [W8] * 5
    lim = np.sqrt(6.0 / (2 * W8))
    W = [_rand_bf16(rs, (W8, W8), -lim, lim) for _ in range(L)]
    B = [(rs.randn(1, W8) * 0.05).astype(np.float32) for _ in range(L)]
    x = _rand_bf16(rs, (M8, W8), 0.0, 1.0)
    trainer = MLPTrainer(widths, M8, loss="mse", optimizer="adam", lr=1e-3, dtype="bfloat16")
    assert trainer.n_params == 268468224
    trainer.set_parameters([{"w": W[i], "b": B[i]} for i in range(L)])
    x16 = bf16.to_bf16(x)
    loss = float(trainer.step(x16, x16))

    # ---- (a) float64 host model with bf16 rounding at the device's storage points
    r16 = lambda a: bf16.round_to_bf16(np.asarray(a, dtype=np.float32)).astype(np.float64)      # noqa: E731
    W64 = [w.astype(np.float64) for w in W]
    acts, masks = [x.astype(np.float64)], []
    for l in range(L):
        z = acts[-1] @ W64[l] + B[l]
        if l < L - 1:
            masks.append(z >= 0)
            acts.append(r16(np.maximum(z, 0)))
        else:
            acts.append(r16(z))
    err = acts[-1] - x
    ref_loss = float((err ** 2).sum() / M8)
    np.testing.assert_allclose(loss, ref_loss, rtol=2e-3)         # bf16 outputs: |pred| rounding flips near ties only
    dz = r16(np.float32(2.0 / M8) * err.astype(np.float32))
    in_rows = np.sort(rs.choice(W8, 16, replace=False))
    worst = 0.0
    for l in reversed(range(L)):
        gw = np.asarray(trainer.grad_view(l, "w"))[in_rows].astype(np.float64)
        ref = acts[l][:, in_rows].T @ dz
        rel = np.linalg.norm(gw - ref) / np.linalg.norm(ref)
        worst = max(worst, rel)
        assert rel <= 5e-3, ("dW%d sampled rows, relative L2 error vs the bf16-faithful model" % l, rel)
        gb = np.asarray(trainer.grad_view(l, "b"), dtype=np.float64)
        refb = dz.sum(0, keepdims=True)
        assert np.linalg.norm(gb - refb) / np.linalg.norm(refb) <= 5e-3, "db%d" % l
        if l > 0:
            dz = r16((dz @ W64[l].T) * masks[l - 1])

    # ---- (b) oracle/closed_form.py, pure float64 (no bf16 rounding anywhere): the loose bars of the 512-wide test
    oracle = ClosedFormMLP(W, B, loss="mse", optimizer="adam", lr=1e-3)
    ref_loss64, _, gW, gb = oracle.loss_and_grads(x, x)
    np.testing.assert_allclose(loss, ref_loss64, rtol=2e-2)
    for l in range(L):
        gw = np.asarray(trainer.grad_view(l, "w"))[in_rows].astype(np.float64)
        rel = np.linalg.norm(gw - gW[l][in_rows]) / np.linalg.norm(gW[l][in_rows])
        assert rel <= 3e-2, ("dW%d vs closed form" % l, rel)

    # ---- the update: fp32 master weights moved by Adam's first step (|step| = lr wherever g != 0), the bf16 working
    # copy and its transpose (what the next forward reads) follow the master copy
    for l in (0, L - 1):
        p = np.asarray(trainer.param_view(l, "w"))
        moved = np.abs(p - W[l])
        assert moved.max() <= 1.001e-3 and np.median(moved) > 0.9e-3
    # the second step must read the REFRESHED bf16 W / W^T copies: its loss against the bf16-faithful host forward on
    # the device's own updated master weights (rounded to bf16 like the working copies are).  (With Adam at lr = 1e-3
    # all 8192 inputs of a unit move coherently, so this net's loss explodes after one step — in float64 just the same;
    # the check is that the device computes THAT loss, not that the loss falls.)
    W1 = [bf16.round_to_bf16(np.asarray(trainer.param_view(l, "w"))).astype(np.float64) for l in range(L)]
    B1 = [np.asarray(trainer.param_view(l, "b"), dtype=np.float64) for l in range(L)]
    a = x.astype(np.float64)
    for l in range(L):
        z = a @ W1[l] + B1[l]
        a = r16(np.maximum(z, 0)) if l < L - 1 else r16(z)
    ref_loss2 = float(((a - x) ** 2).sum() / M8)
    loss2 = float(trainer.step(x16, x16))
    np.testing.assert_allclose(loss2, ref_loss2, rtol=2e-3)
    assert abs(loss2 - loss) > 0.5 * loss                         # a stale W^T copy would reproduce the first loss

    # ---- the same two steps with the weight gradients consumed in the dW epilogues and never stored (what bench.py times
    # for this config): losses, master weights and Adam state bit-identical to the run above
    p_ref = np.asarray(trainer.params).copy()
    m_ref = np.asarray(trainer.adam_m).copy()
    del trainer
    t2 = MLPTrainer(widths, M8, loss="mse", optimizer="adam", lr=1e-3, dtype="bfloat16").keep_grads(False)
    t2.set_parameters([{"w": W[i], "b": B[i]} for i in range(L)])
    assert float(t2.step(x16, x16)) == loss and float(t2.step(x16, x16)) == loss2
    assert np.array_equal(np.asarray(t2.params), p_ref) and np.array_equal(np.asarray(t2.adam_m), m_ref)
    # ... in 4 L + 1 = 17 launches: 4 forward GEMMs, the prep launch (loss, dz, dz^T, beta powers), per layer one launch for both
    # transposed operands of its dW product + the dX GEMM + the dW GEMM with Adam in the epilogue, one launch for the four biases
    # (25 launches before; DESIGN §5 for why the transposes stay launches)
    import ctypes
    n = ctypes.c_int(0)
    t2._lib.mlp_launch_window(t2._h, 0, -1, ctypes.byref(n))
    assert n.value == 4 * L + 1 == 17, n.value
    del t2
    # ... and the 13-launch form (transposed operands from the producing GEMMs' epilogues; kept for the A/B probe) gives the same bits
    os.environ["TNN_E_STEP"] = "ct"
    try:
        t3 = MLPTrainer(widths, M8, loss="mse", optimizer="adam", lr=1e-3, dtype="bfloat16").keep_grads(False)
        t3.set_parameters([{"w": W[i], "b": B[i]} for i in range(L)])
        assert float(t3.step(x16, x16)) == loss and float(t3.step(x16, x16)) == loss2
        assert np.array_equal(np.asarray(t3.params), p_ref) and np.array_equal(np.asarray(t3.adam_m), m_ref)
        t3._lib.mlp_launch_window(t3._h, 0, -1, ctypes.byref(n))
        assert n.value == 3 * L + 1 == 13, n.value
    finally:
        os.environ.pop("TNN_E_STEP", None)
