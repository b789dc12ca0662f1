"""bf16 path on the MI355X: K-contiguous MFMA GEMM (fp32 accumulate) against float64 numpy on bf16-rounded
operands, epilogues, transpose/cast/colsum, and the master-weight Adam."""

import numpy as np
import pytest

import tinynn_autograd_amd as tn
from tinynn_autograd_amd import bf16


@pytest.mark.gpu
def test_bf16_gemm_nt_and_epilogues():
    rs = np.random.RandomState(41)
    for (M, N, K) in ((512, 1024, 2048), (128, 128, 64), (200, 330, 192), (1000, 77, 512)):
        a = bf16.round_to_bf16(rs.uniform(-1, 1, (M, K)).astype(np.float32))
        b = bf16.round_to_bf16(rs.uniform(-1, 1, (N, K)).astype(np.float32))
        ref = a.astype(np.float64) @ b.astype(np.float64).T
        bound = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
        A, B = bf16.to_bf16(a), bf16.to_bf16(b)
        assert np.array_equal(np.asarray(bf16.to_f32(A)), a)                # bf16 values survive the round trip
        c = np.asarray(bf16.gemm_nt(A, B), dtype=np.float64)
        assert (np.abs(c - ref) <= 2e-6 * bound + 1e-30).all(), (M, N, K, np.abs(c - ref).max())   # fp32 accumulation
        # bf16 output: one extra rounding (2^-9 relative)
        c16 = np.asarray(bf16.to_f32(bf16.gemm_nt(A, B, out_dtype=np.uint16)), dtype=np.float64)
        assert (np.abs(c16 - ref) <= 4e-3 * np.abs(ref) + 2e-6 * bound).all()
        # bias + ReLU with the mask kept in the sign bit, then the mask epilogue of the next product
        bias = rs.randn(N).astype(np.float32)
        y16 = bf16.gemm_nt(A, B, out_dtype=np.uint16, bias=tn.asarray(bias), relu=True, relu_sign=True)
        y = np.asarray(bf16.to_f32(y16))
        z = ref + bias
        assert (np.abs(y - np.maximum(z, 0)) <= 4e-3 * np.abs(z) + 1e-5 * bound + 1e-6).all()
        raw = np.asarray(y16)
        clearly_neg, clearly_pos = z < -1e-3 * bound - 1e-6, z > 1e-3 * bound + 1e-6
        assert ((raw[clearly_neg] & 0x8000) != 0).all() and ((raw[clearly_pos] & 0x8000) == 0).all()
        masked = np.asarray(bf16.gemm_nt(A, B, mask=y16), dtype=np.float64)
        expect = np.where((raw & 0x8000) != 0, 0.0, ref)
        assert (np.abs(masked - expect) <= 2e-6 * bound + 1e-30).all()


@pytest.mark.gpu
def test_bf16_skinny_gemm_split_k_path():
    """Shapes that take the 256 x 128 split-K kernel (tnn_gemm_bf16_sk.h: bf16 output, about one workgroup per CU): ragged
    M / N (clamped edge rows, the unstaged epilogue), a leading dimension that is not a multiple of 8, every epilogue, and
    two different products back to back — a slab or a counter surviving from the previous launch would show at once.
    Results are bit-identical run to run (fixed summation order of the two K slices)."""
    rs = np.random.RandomState(43)
    for (M, N, K) in ((512, 8192, 1024), (300, 8000, 1024), (512, 7990, 640), (257, 16000, 768)):
        a = bf16.round_to_bf16(rs.uniform(-1, 1, (M, K)).astype(np.float32))
        a2 = bf16.round_to_bf16(rs.uniform(-1, 1, (M, K)).astype(np.float32))
        b = bf16.round_to_bf16(rs.uniform(-1, 1, (N, K)).astype(np.float32))
        A, A2, B = bf16.to_bf16(a), bf16.to_bf16(a2), bf16.to_bf16(b)
        b64 = b.astype(np.float64)
        outs = []
        for rep in range(3):
            for src, src16 in ((a, A), (a2, A2)):
                ref = src.astype(np.float64) @ b64.T
                bound = np.abs(src).astype(np.float64) @ np.abs(b64).T
                out16 = bf16.gemm_nt(src16, B, out_dtype=np.uint16)
                raw = np.asarray(out16)
                c16 = np.asarray(bf16.to_f32(out16), dtype=np.float64)
                assert (np.abs(c16 - ref) <= 4e-3 * np.abs(ref) + 2e-6 * bound).all(), (M, N, K, rep)
                outs.append(raw)
        assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[0], outs[4])          # bit-identical run to run
        assert np.array_equal(outs[1], outs[3]) and np.array_equal(outs[1], outs[5])
        ref = a.astype(np.float64) @ b64.T
        bound = np.abs(a).astype(np.float64) @ np.abs(b64).T
        bias = rs.randn(N).astype(np.float32)
        y16 = bf16.gemm_nt(A, B, out_dtype=np.uint16, bias=tn.asarray(bias), relu=True, relu_sign=True)
        y = np.asarray(bf16.to_f32(y16))
        z = ref + bias
        assert (np.abs(y - np.maximum(z, 0)) <= 4e-3 * np.abs(z) + 1e-5 * bound + 1e-6).all()
        raw = np.asarray(y16)
        clearly_neg, clearly_pos = z < -1e-3 * bound - 1e-6, z > 1e-3 * bound + 1e-6
        assert ((raw[clearly_neg] & 0x8000) != 0).all() and ((raw[clearly_pos] & 0x8000) == 0).all()
        masked16 = bf16.gemm_nt(A, B, out_dtype=np.uint16, mask=y16)
        masked = np.asarray(bf16.to_f32(masked16), dtype=np.float64)
        expect = np.where((raw & 0x8000) != 0, 0.0, ref)
        assert (np.abs(masked - expect) <= 4e-3 * np.abs(expect) + 2e-6 * bound).all()
        assert (masked[(raw & 0x8000) != 0] == 0).all()


@pytest.mark.gpu
def test_bf16_gemm_transposed_second_output():
    """tnn_gemm_bf16_nt_t: C and C^T from ONE launch (the split-K kernel's LDS image read a second time column-wise) — C bit-equal
    to tnn_gemm_bf16_nt's, C^T its exact transpose, for the plain / bias + ReLU / mask epilogues (the mask must be applied to
    the transposed copy too), interior and ragged tiles, and on a shape the split-K kernel does not take (GEMM + transpose)."""
    rs = np.random.RandomState(44)
    for (M, N, K) in ((512, 8192, 1024), (512, 8064, 640), (300, 8000, 1024), (256, 384, 128)):
        a = bf16.round_to_bf16(rs.uniform(-1, 1, (M, K)).astype(np.float32))
        b = bf16.round_to_bf16(rs.uniform(-1, 1, (N, K)).astype(np.float32))
        A, B = bf16.to_bf16(a), bf16.to_bf16(b)
        bias = tn.asarray(rs.randn(N).astype(np.float32))
        plain = np.asarray(bf16.gemm_nt(A, B, out_dtype=np.uint16))
        c, ct = bf16.gemm_nt_t(A, B)
        assert np.array_equal(np.asarray(c), plain) and np.array_equal(np.asarray(ct), plain.T), (M, N, K, "plain")
        act16 = bf16.gemm_nt(A, B, out_dtype=np.uint16, bias=bias, relu=True, relu_sign=True)
        c, ct = bf16.gemm_nt_t(A, B, bias=bias, relu=True, relu_sign=True)
        assert np.array_equal(np.asarray(c), np.asarray(act16)) and np.array_equal(np.asarray(ct), np.asarray(act16).T), (M, N, K, "relu")
        masked = np.asarray(bf16.gemm_nt(A, B, out_dtype=np.uint16, mask=act16))
        c, ct = bf16.gemm_nt_t(A, B, mask=act16)
        assert np.array_equal(np.asarray(c), masked) and np.array_equal(np.asarray(ct), masked.T), (M, N, K, "mask")
        assert (masked[(np.asarray(act16) & 0x8000) != 0] & 0x7fff == 0).all()


@pytest.mark.gpu
def test_bf16_prep_launch_and_multi_layer_bias_launch():
    """tnn_mse_bf16_prep (loss + dz + dz^T + x^T + beta powers in one launch) against tnn_mse_bf16_tick + two transposes, and
    tnn_bias_bf16_adam_multi (every layer's bias gradient + Adam in one launch) against one tnn_bias_bf16_adam per layer:
    every output bit-identical (the loss to float32 rounding of an f64 sum)."""
    import ctypes
    from tinynn_autograd_amd import _lib
    lib = _lib.get()
    rs = np.random.RandomState(45)
    rows, cols, xc = 256, 640, 384
    pred = bf16.to_bf16(rs.randn(rows, cols).astype(np.float32))
    y = bf16.to_bf16(rs.randn(rows, cols).astype(np.float32))
    x = bf16.to_bf16(rs.rand(rows, xc).astype(np.float32))
    dz_ref, loss_ref = tn.empty((rows, cols), np.uint16), tn.empty((2,), np.float32)
    pows_ref = tn.asarray(np.array([0.9, 0.999, 0.0, 0.0]), dtype=np.float64)
    lib.mse_bf16_tick(pred._ptr, y._ptr, rows * cols, rows, loss_ref._ptr, loss_ref._ptr + 4, dz_ref._ptr, pows_ref._ptr, 0.9, 0.999)
    dz, dzt, xt = tn.empty((rows, cols), np.uint16), tn.empty((cols, rows), np.uint16), tn.empty((xc, rows), np.uint16)
    loss = tn.empty((2,), np.float32)
    pows = tn.asarray(np.array([0.9, 0.999, 0.0, 0.0]), dtype=np.float64)
    ws, ticket = tn.empty((rows // 64 * (cols // 64),), np.float64), tn.asarray(np.zeros(32, np.int64))
    for rep in range(2):                                     # the ticket returns to zero: a second launch works the same
        lib.mse_bf16_prep(pred._ptr, y._ptr, rows, cols, rows, loss._ptr, loss._ptr + 4, dz._ptr, dzt._ptr, x._ptr, xc, xt._ptr,
                          ws._ptr, ticket._ptr, pows._ptr if rep == 0 else None, 0.9, 0.999)
        assert np.array_equal(np.asarray(dz), np.asarray(dz_ref)) and np.array_equal(np.asarray(dzt), np.asarray(dz_ref).T)
        assert np.array_equal(np.asarray(xt), np.asarray(x).T)
        np.testing.assert_allclose(np.asarray(loss), np.asarray(loss_ref), rtol=2e-7)
        assert np.asarray(loss)[0] == np.asarray(loss)[1] and not np.asarray(ticket).any()
    assert np.array_equal(np.asarray(pows), np.asarray(pows_ref))

    # the biases of several layers in ONE launch against one tnn_bias_bf16_adam launch per layer
    import ctypes
    widths = [640, 200, 64]
    dzs = [bf16.to_bf16((rs.randn(rows, c) * 0.01).astype(np.float32)) for c in widths]
    results = []
    for multi in (False, True):
        st = [[tn.asarray(np.random.RandomState(8 + i).randn(c).astype(np.float32)), tn.zeros((c,), np.float32), tn.zeros((c,), np.float32),
               tn.empty((c,), np.uint16), tn.empty((c,), np.float32)] for i, c in enumerate(widths)]          # p, m, v, w16, db
        pw = tn.asarray(np.array([0.9, 0.999, 0.0, 0.0]), dtype=np.float64)
        for _ in range(2):
            if multi:
                arr = lambda k: (ctypes.c_void_p * 3)(*[s_[k]._ptr for s_ in st])          # noqa: E731
                lib.bias_bf16_adam_multi(3, (ctypes.c_void_p * 3)(*[d._ptr for d in dzs]), rows, (ctypes.c_int64 * 3)(*widths),
                                         arr(4), arr(0), arr(1), arr(2), arr(3), 1e-3, 0.9, 0.999, 1e-8, pw._ptr)
            else:
                for d, c, s_ in zip(dzs, widths, st):
                    lib.bias_bf16_adam(d._ptr, rows, c, s_[4]._ptr, s_[0]._ptr, s_[1]._ptr, s_[2]._ptr, s_[3]._ptr, 1e-3, 0.9, 0.999,
                                       1e-8, pw._ptr)
        results.append([np.asarray(a).copy() for s_ in st for a in s_])
    for i, (a, b) in enumerate(zip(*results)):
        assert np.array_equal(a, b), i
    np.testing.assert_allclose(results[1][4], np.asarray(bf16.to_f32(dzs[0])).astype(np.float64).sum(0), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_bf16_two_transposes_in_one_launch():
    from tinynn_autograd_amd import _lib
    lib = _lib.get()
    rs = np.random.RandomState(46)
    for (r1, c1, r2, c2) in ((512, 8192, 512, 8192), (128, 256, 128, 384), (64, 64, 320, 192), (36, 20, 100, 52), (30, 7, 64, 64)):
        a, b = bf16.to_bf16(rs.randn(r1, c1).astype(np.float32)), bf16.to_bf16(rs.randn(r2, c2).astype(np.float32))
        at_, bt_ = tn.empty((c1, r1), np.uint16), tn.empty((c2, r2), np.uint16)
        lib.transpose2_bf16(a._ptr, at_._ptr, r1, c1, b._ptr, bt_._ptr, r2, c2)
        assert np.array_equal(np.asarray(at_), np.asarray(a).T) and np.array_equal(np.asarray(bt_), np.asarray(b).T), (r1, c1, r2, c2)


@pytest.mark.gpu
def test_bf16_transpose_colsum_mse_adam():
    from tinynn_autograd_amd import _lib
    rs = np.random.RandomState(42)
    x = bf16.round_to_bf16(rs.randn(300, 130).astype(np.float32))
    X = bf16.to_bf16(x)
    assert np.array_equal(np.asarray(bf16.to_f32(bf16.transpose(X))), x.T)
    for shape in ((512, 8192), (64, 64), (132, 260), (36, 8), (129, 70)):            # fast (x4) and fallback shapes
        z = bf16.round_to_bf16(rs.randn(*shape).astype(np.float32))
        assert np.array_equal(np.asarray(bf16.to_f32(bf16.transpose(bf16.to_bf16(z)))), z.T), shape
    cs = tn.empty((130,))
    _lib.get().colsum_bf16(X._ptr, cs._ptr, 300, 130)
    np.testing.assert_allclose(np.asarray(cs), x.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    big = bf16.round_to_bf16(rs.randn(2048, 256).astype(np.float32))
    cs2 = tn.empty((256,))
    _lib.get().colsum_bf16(bf16.to_bf16(big)._ptr, cs2._ptr, 2048, 256)
    np.testing.assert_allclose(np.asarray(cs2), big.astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    y = bf16.round_to_bf16(rs.randn(300, 130).astype(np.float32))
    loss, d = tn.empty(()), tn.empty((300, 130), np.uint16)
    _lib.get().mse_bf16(X._ptr, bf16.to_bf16(y)._ptr, 300 * 130, 300, loss._ptr, d._ptr)
    e = x.astype(np.float64) - y
    np.testing.assert_allclose(float(loss), (e ** 2).sum() / 300, rtol=1e-6)
    np.testing.assert_allclose(np.asarray(bf16.to_f32(d)), 2 * e / 300, rtol=8e-3, atol=1e-7)
    n = 10007
    p0 = rs.randn(n).astype(np.float32); g = (rs.randn(n) * 1e-2).astype(np.float32)
    P, G, M_, V_ = tn.asarray(p0), tn.asarray(g), tn.zeros((n,)), tn.zeros((n,))
    W16 = tn.empty((n,), np.uint16)
    pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
    _lib.get().adam_master_bf16(P._ptr, G._ptr, M_._ptr, V_._ptr, W16._ptr, n, 1e-3, 0.9, 0.999, 1e-8, pows._ptr)
    m = 0.1 * g.astype(np.float64); v = 0.001 * g.astype(np.float64) ** 2
    ref = p0 - 1e-3 * (m / 0.1) / (np.sqrt(v / 0.001) + 1e-8)
    np.testing.assert_allclose(np.asarray(P), ref, rtol=0, atol=2e-6)
    assert np.array_equal(np.asarray(bf16.to_f32(W16)), bf16.round_to_bf16(np.asarray(P)))
    # the tiled variant that also emits the transposed working copy: bit-identical to the flat kernel
    for (R, C) in ((192, 136), (64, 64), (100, 36), (1, 40), (30, 10)):
        p0 = rs.randn(R, C).astype(np.float32); g = (rs.randn(R, C) * 1e-2).astype(np.float32)
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
            out.append([np.asarray(t) for t in (P, M_, V_, W16)] + [np.asarray(pows)])
            if tiled:
                assert np.array_equal(np.asarray(WT16), np.asarray(W16).T), (R, C)
        for a, b in zip(*out):
            assert np.array_equal(a, b), (R, C)


@pytest.mark.gpu
def test_bf16_trainer_tracks_float64_oracle():
    """4-layer 512-wide ReLU MLP, sum-of-squares loss, Adam, bf16 storage / fp32 master weights: loss and
    gradients against the float64 closed form evaluated on the SAME bf16-rounded weights and inputs; bf16
    activations and dz add ~2^-9 relative noise per tensor, hence the 2e-2 bars."""
    from oracle.closed_form import ClosedFormMLP
    from tinynn_autograd_amd.fused import MLPTrainer
    rs = np.random.RandomState(43)
    widths, m = [512, 512, 512, 512, 512], 128
    a = np.sqrt(6.0 / 1024)
    W = [bf16.round_to_bf16(rs.uniform(-a, a, (512, 512)).astype(np.float32)) for _ in range(4)]
    B = [np.zeros((1, 512), np.float32) for _ in range(4)]
    x = bf16.round_to_bf16(rs.rand(m, 512).astype(np.float32))
    trainer = MLPTrainer(widths, m, loss="mse", optimizer="adam", lr=1e-3, dtype="bfloat16")
    trainer.set_parameters([{"w": W[i], "b": B[i]} for i in range(4)])
    oracle = ClosedFormMLP(W, B, loss="mse", optimizer="adam", lr=1e-3)
    x16 = bf16.to_bf16(x)
    pred = np.asarray(trainer.forward(x16), dtype=np.float64)
    acts, _ = oracle.forward(x)
    assert np.abs(pred - acts[-1]).max() <= 2e-2 * np.abs(acts[-1]).max()
    losses, ref_losses = [], []
    for step in range(3):
        ref_loss, _, gW, gb = oracle.loss_and_grads(x, x)
        loss = float(trainer.step(x16, x16))
        losses.append(loss); ref_losses.append(ref_loss)
        if step == 0:
            for l in range(4):
                g = np.asarray(trainer.grad_view(l, "w"), dtype=np.float64)
                rel = np.linalg.norm(g - gW[l]) / np.linalg.norm(gW[l])
                assert rel <= 3e-2, ("dW%d relative L2 error" % l, rel)
                gb_d = np.asarray(trainer.grad_view(l, "b"), dtype=np.float64)
                assert np.linalg.norm(gb_d - gb[l]) / np.linalg.norm(gb[l]) <= 3e-2
        oracle.step(x, x)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-2)
    assert losses[2] < losses[0]                              # it trains
    # the fp32 master copy moved by Adam, and the bf16 working copy is its rounding
    p = np.asarray(trainer.param_view(0, "w"))
    assert np.abs(p - W[0]).max() > 1e-4


@pytest.mark.gpu
def test_bf16_dw_gemm_with_adam_epilogue_equals_gemm_then_adam():
    """tnn_gemm_bf16_nt_adam (Adam consumes the weight gradient in the dW GEMM's epilogue) against the two launches it
    replaces — tnn_gemm_bf16_nt (fp32 out) + tnn_adam_master_bf16_2d: master weights, both moments, the bf16 working copy
    and its transpose bit-identical over two steps; full, ragged and tiny shapes; with and without the stored gradient."""
    from tinynn_autograd_amd import _lib
    lib = _lib.get()
    rs = np.random.RandomState(47)
    for (M, N, K) in ((512, 384, 128), (256, 256, 512), (200, 72, 64), (128, 1000, 192), (4, 8, 64)):
        a = bf16.to_bf16(bf16.round_to_bf16(rs.uniform(-1, 1, (M, K)).astype(np.float32)))
        b = bf16.to_bf16(bf16.round_to_bf16((rs.uniform(-1, 1, (N, K)) * 1e-2).astype(np.float32)))
        p0 = rs.randn(M, N).astype(np.float32)
        res = []
        for fused, keep in ((False, True), (True, True), (True, False)):
            P, M_, V_ = tn.asarray(p0), tn.zeros((M, N)), tn.zeros((M, N))
            G = tn.asarray(np.full((M, N), 7.0, np.float32))
            W16, WT16 = tn.empty((M, N), np.uint16), tn.empty((N, M), np.uint16)
            pows = tn.asarray(np.array([1.0, 1.0, 0, 0]), dtype=np.float64)
            for _ in range(2):
                if fused:
                    lib.adam_tick(pows._ptr, 0.9, 0.999)
                    lib.gemm_bf16_nt_adam(M, N, K, a._ptr, K, b._ptr, K, G._ptr if keep else None, P._ptr, M_._ptr, V_._ptr,
                                          W16._ptr, WT16._ptr, 1e-3, 0.9, 0.999, 1e-8, pows._ptr)
                else:
                    lib.gemm_bf16_nt(M, N, K, a._ptr, K, b._ptr, K, G._ptr, N, _lib.F32, None, 0, 0, None, 0)
                    lib.adam_master_bf16_2d(P._ptr, G._ptr, M_._ptr, V_._ptr, W16._ptr, WT16._ptr, M, N,
                                            1e-3, 0.9, 0.999, 1e-8, pows._ptr, 1)
            res.append([np.asarray(t) for t in (P, M_, V_, W16, WT16, pows, G)])
        for got in res[1:]:
            for name, x0, x1 in zip(("p", "m", "v", "w16", "wT16", "pows"), res[0], got):
                assert np.array_equal(x0, x1), (name, M, N, K)
        assert np.array_equal(res[0][6], res[1][6])                      # the stored gradient, when asked for
        assert (res[2][6] == 7.0).all()                                  # and untouched when not
        assert np.array_equal(res[0][4], res[0][3].T) and np.abs(res[0][0] - p0).max() > 5e-4


@pytest.mark.gpu
def test_bf16_trainer_fused_step_equals_separate_launches(monkeypatch):
    """The bf16 trainer's `step` against its forward / backward / update entry points: identical losses, parameters and
    Adam state after three steps — with the weight gradients stored (default: the same launches) and with
    keep_grads(False), where Adam consumes each weight gradient in the epilogue of the dW GEMM."""
    from tinynn_autograd_amd.fused import MLPTrainer
    rs = np.random.RandomState(48)
    widths, m = [256, 384, 128, 256], 128
    W = [bf16.round_to_bf16(rs.uniform(-0.08, 0.08, (widths[i], widths[i + 1])).astype(np.float32)) for i in range(3)]
    B = [(rs.randn(1, widths[i + 1]) * 0.05).astype(np.float32) for i in range(3)]
    x16 = bf16.to_bf16(bf16.round_to_bf16(rs.rand(m, 256).astype(np.float32)))
    runs = []
    for mode in ("separate", "step", "step_nokeep", "step_nokeep_ct", "step_nokeep_long"):
        t = MLPTrainer(widths, m, loss="mse", optimizer="adam", lr=1e-3, dtype="bfloat16")
        t.set_parameters([{"w": W[i], "b": B[i]} for i in range(3)])
        if mode.startswith("step_nokeep"):
            t.keep_grads(False)                                          # -> Adam in the dW epilogues
            # the three launch sequences of that step (tnn_mlp.cpp mlp16_step_form): 4L + 1 launches (default), 3L + 1 with the
            # transposed operands from the GEMM epilogues, the long sequence
            monkeypatch.delenv("TNN_E_STEP", raising=False)
            if mode != "step_nokeep":
                monkeypatch.setenv("TNN_E_STEP", mode.rsplit("_", 1)[1])
        losses = []
        for _ in range(3):
            if mode == "separate":
                t._lib.mlp_forward_stats(t._h, x16._ptr, m, None)
                t._lib.mlp_backward(t._h, x16._ptr, x16._ptr, m, m, None, None)
                t._lib.mlp_update(t._h)
                losses.append(float(t.loss_slot))
            else:
                losses.append(float(t.step(x16, x16)))
        runs.append((losses, np.asarray(t.params).copy(), np.asarray(t.adam_m).copy(), np.asarray(t.adam_v).copy(),
                     [np.asarray(t.grad_view(l, "w")).copy() for l in range(3)],
                     np.asarray(t.forward(x16)).copy(), np.asarray(t.weights_bf16()).copy()))
        # the bf16 working copy IS the rounded masters — also W_0's [in, out] copy, which the default step does not write (it
        # has no reader inside a step) and the accessor re-derives
        for l in range(3):
            assert np.array_equal(np.asarray(t.weights_bf16(l)), np.asarray(bf16.to_bf16(np.asarray(t.param_view(l, "w"))))), (mode, l)
    monkeypatch.delenv("TNN_E_STEP", raising=False)
    for got in runs[1:]:
        assert got[0] == runs[0][0]
        for k in (1, 2, 3, 5, 6):
            assert np.array_equal(got[k], runs[0][k]), k
    for l in range(3):
        assert np.array_equal(runs[1][4][l], runs[0][4][l])              # stored gradients of the default step


@pytest.mark.gpu
def test_bias_launch_and_bf16_gradient_adam_vs_numpy():
    """tnn_bias_bf16_adam (db = column sums of a bf16 dz + Adam on the fp32 master bias + its bf16 copy, one launch) and
    tnn_adam_master_g16 (Adam on a slice whose gradient arrives as bf16: the reduce-scattered slice of the sharded-optimizer
    step) against float64 numpy; ragged column counts and an unaligned view take the element-wise path."""
    from tinynn_autograd_amd import _lib
    lib = _lib.get()
    rs = np.random.RandomState(77)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    for rows, cols in ((512, 8192), (100, 70), (64, 4)):
        dz = bf16.round_to_bf16((rs.randn(rows, cols) * 1e-2).astype(np.float32))
        DZ = bf16.to_bf16(dz)
        p0, m0, v0 = rs.randn(cols).astype(np.float32), (rs.randn(cols) * 1e-3).astype(np.float32), (rs.rand(cols) * 1e-5).astype(np.float32)
        P, M, V, DB = tn.asarray(p0), tn.asarray(m0), tn.asarray(v0), tn.zeros((cols,))
        W16 = tn.empty((cols,), np.uint16)
        pows = tn.asarray(np.array([b1 ** 3, b2 ** 3, 0, 0]), dtype=np.float64)
        lib.bias_bf16_adam(DZ._ptr, rows, cols, DB._ptr, P._ptr, M._ptr, V._ptr, W16._ptr, lr, b1, b2, eps, pows._ptr)
        g = dz.astype(np.float64).sum(0)
        np.testing.assert_allclose(np.asarray(DB), g, rtol=0, atol=2e-6 * np.abs(dz).sum(0).max())
        gd = np.asarray(DB, dtype=np.float64)                        # the update is checked on the device's own sums
        ob1, ob2 = float(np.float32(1) - np.float32(b1)), float(np.float32(1) - np.float32(b2))   # the kernels' float32 1 - beta
        m1 = m0 + ob1 * (gd - m0)
        v1 = v0 + ob2 * (gd * gd - v0)
        p1 = p0 - lr * (m1 / (1 - b1 ** 3)) / (np.sqrt(v1 / (1 - b2 ** 3)) + eps)
        np.testing.assert_allclose(np.asarray(M), m1, rtol=0, atol=2e-6 * np.abs(m1).max())
        np.testing.assert_allclose(np.asarray(V), v1, rtol=0, atol=2e-6 * np.abs(v1).max())
        np.testing.assert_allclose(np.asarray(P), p1, rtol=0, atol=2e-6 * np.abs(p1).max())
        assert np.array_equal(np.asarray(bf16.to_f32(W16)), bf16.round_to_bf16(np.asarray(P)))
    for n in (8192 * 16, 1000, 7):
        g = bf16.round_to_bf16((rs.randn(n) * 1e-2).astype(np.float32))
        p0, m0, v0 = rs.randn(n).astype(np.float32), (rs.randn(n) * 1e-3).astype(np.float32), (rs.rand(n) * 1e-5).astype(np.float32)
        P, M, V, G16, W16 = tn.asarray(p0), tn.asarray(m0), tn.asarray(v0), bf16.to_bf16(g), tn.empty((n,), np.uint16)
        pows = tn.asarray(np.array([b1 ** 2, b2 ** 2, 0, 0]), dtype=np.float64)
        lib.adam_master_g16(P._ptr, G16._ptr, M._ptr, V._ptr, W16._ptr, n, lr, b1, b2, eps, pows._ptr)
        gd = g.astype(np.float64)
        m1 = m0 + ob1 * (gd - m0)
        v1 = v0 + ob2 * (gd * gd - v0)
        p1 = p0 - lr * (m1 / (1 - b1 ** 2)) / (np.sqrt(v1 / (1 - b2 ** 2)) + eps)
        np.testing.assert_allclose(np.asarray(M), m1, rtol=0, atol=2e-6 * np.abs(m1).max())
        np.testing.assert_allclose(np.asarray(V), v1, rtol=0, atol=2e-6 * np.abs(v1).max())
        np.testing.assert_allclose(np.asarray(P), p1, rtol=0, atol=2e-6 * np.abs(p1).max())
        assert np.array_equal(np.asarray(bf16.to_f32(W16)), bf16.round_to_bf16(np.asarray(P)))
