// Whole-step Dense-MLP trainer: the loop body of examples/mnist/run.py:79-83
// (zero_grad -> forward -> loss -> backward -> step) as a fixed sequence of launches on
// device-resident state.  Host-only C++ on top of the C-ABI primitives, so the same file is linked
// into libtnn_hip.so (HIP kernels) and into the CPU test twin.
//
// State layout (HBM): four flat arenas  params | grads | m | v , each n_params(+pad) elements, in
// the order core/optimizer.py:14-15 flattens them: layer by layer, "w" [in,out] then "b" [1,out]
// (core/layers.py:34-35).  grads[n_params] is one extra slot that carries this shard's loss so the
// data-parallel all-reduce of the gradient arena also sums the loss for free.
//
// Per step (L Dense layers, ReLU between them):
//   forward  : L  x  gemm_bias_act (NN, bias + ReLU fused, ReLU mask kept in the sign bit of 0)
//   head     : mlp_head = last Dense forward + softmax NLL + its backward in one launch (unsharded);
//              sharded: gemm_bias_act | softmax_nll_stats | exchange | softmax_nll_fwd_bwd | dense_bwd
//   (loss    : softmax_nll_fused (sharded: softmax_nll_stats | exchange | softmax_nll_fwd_bwd) or mse_fwd_bwd
//   backward : L  x  dense_bwd (dW = X^T dZ, db = column-sum dZ, dX = (dZ W^T)*mask in one launch when small)
//   update   : 1  x  fused Adam / SGD over the whole arena
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include <stdlib.h>

#include "tnn_hip.h"

namespace tnn {
void set_error(const char* fmt, ...);
}

namespace {

struct Mlp {
    int L = 0;
    std::vector<int64_t> w;          // widths, L+1
    int64_t max_rows = 0;
    int loss_kind = 0, opt_kind = 0, dtype = TNN_F32;
    double lr = 1e-3, b1 = 0.9, b2 = 0.999, eps = 1e-8;
    int64_t n_params = 0, arena = 0;
    size_t esz = 4;
    char *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
    void* pows = nullptr;            // Adam state, double[4]
    void* stats = nullptr;           // {M, S} of the last forward_stats
    bool keep_grads = true;          // tnn_mlp_keep_grads: false lets a step consume weight gradients without storing them
    void* zpart = nullptr;           // [w[L-1] / 16][max_rows][w[L]] partial logits (5-launch step, see tnn_mlp_step)
    void* ticket = nullptr;          // arrival counter of the forward launch whose last workgroup reduces the shard's statistics
    void* stats_all = nullptr;       // [world, 2] gathered shard stats (data-parallel step)
    int stats_all_world = 0;
    std::vector<int64_t> w_off, b_off;
    std::vector<void*> act;          // act[l]  = output of layer l       [max_rows, w[l+1]]
    std::vector<void*> dact;         // dact[l] = dLoss/d(pre-activation) [max_rows, w[l+1]]
    // bf16 mode (dtype == TNN_BF16): the four arenas above stay fp32 (master weights, gradients, Adam state);
    // the GEMMs consume bf16 working copies, every operand K-contiguous (see tnn_gemm_bf16.hip)
    bool bf16 = false;
    char* w16 = nullptr;             // bf16 copy of the whole parameter arena (W_l at w_off[l], [in,out])
    bool w16_first_stale = false;    // a step form that does not refresh W_0's copy has run (or been recorded)
    std::vector<void*> wT16;         // W_l^T [out,in]  (forward operand)
    std::vector<void*> actT16;       // act[l]^T [w[l+1], max_rows]   (dW operand of layer l+1)
    std::vector<void*> dactT16;      // dact[l]^T [w[l+1], max_rows]  (dW operand of layer l)
    void* xT16 = nullptr;            // x^T [w[0], max_rows]
    void* prep_ws = nullptr;         // block partials of the prep launch's loss reduction (tnn_mse_bf16_prep)
    // sharded-optimizer step (data-parallel bf16 trainer, mlp16_step_zero): the weight gradients as bf16 in arena order
    // (wire format of the reduce-scatter) and a contiguous fp32 staging vector [b_0 | ... | b_{L-1} | loss] for the one
    // small all-reduce of the bias gradients and the loss
    char* g16 = nullptr;
    float* bias_g = nullptr;
    std::vector<int64_t> bias_g_off;
    // world size of the last sharded-optimizer step (0: the fp32 master / m / v arenas are whole).  While it is > 1 every
    // rank's masters are current for ITS row slice of each weight matrix only (tnn_mlp_gather_masters makes them whole)
    int masters_world = 0;
    // measurement hook (tnn_mlp_launch_window): primitive calls of a step are numbered 0, 1, ... in issue order and only
    // those inside [win_lo, win_hi) are executed, so each launch of the step can be replayed and timed on its own
    // No window set (the default): every call runs and the counter only reports launches per step; it is reset at the
    // top of every public entry point, so it never grows past one step's worth of calls.
    bool windowed = false;
    int win_lo = 0, win_hi = 0;
    int64_t call_idx = 0;
};

#define MLP_TRY(call)            \
    do {                         \
        int rc__ = (call);       \
        if (rc__) return rc__;   \
    } while (0)

// one primitive call (= one launch for the MNIST-size step) of a training step, subject to the launch window
#define STEP_CALL(h, call)                                              \
    do {                                                                                        \
        const int64_t idx__ = (h)->call_idx++;                                                   \
        if (!(h)->windowed || (idx__ >= (h)->win_lo && idx__ < (h)->win_hi)) MLP_TRY(call);     \
    } while (0)

inline void* at(char* base, int64_t elem_off, size_t esz) { return base + elem_off * (int64_t)esz; }

int mlp_forward(Mlp* h, const void* x, int64_t rows, int n_layers = -1) {
    const void* in = x;
    if (n_layers < 0) n_layers = h->L;
    for (int l = 0; l < n_layers; ++l) {
        bool hidden = l < h->L - 1;
        STEP_CALL(h, tnn_gemm_bias_act(0, 0, rows, h->w[l + 1], h->w[l], in, h->w[l],
                                       at(h->params, h->w_off[l], h->esz), h->w[l + 1],
                                       at(h->params, h->b_off[l], h->esz),
                                       hidden ? TNN_ACT_RELU : TNN_ACT_NONE, hidden ? 1 : 0, h->act[l],
                                       h->w[l + 1], h->dtype));
        in = h->act[l];
    }
    return 0;
}

// Layer l's gradients (W_l then b_l, contiguous in the arena; the last layer's bucket also carries the loss slot
// behind the arena) go to the communication stream as soon as their backward launch is enqueued.
int allreduce_layer_bucket(Mlp* h, int l) {
    const int64_t first = h->w_off[l];
    int64_t count = h->w[l] * h->w[l + 1] + h->w[l + 1];
    if (l == h->L - 1) count = h->n_params + 1 - first;
    return tnn_allreduce_async(at(h->grads, first, h->esz), count, h->bf16 ? TNN_F32 : h->dtype, TNN_RSUM);
}

// gradients of every layer from dact[L-1] (set by the loss kernel) down to layer 0
// bucket: data-parallel step of a large net — all-reduce each layer's gradients as soon as they are enqueued
int mlp_backward_layers(Mlp* h, const void* x, int64_t rows, int from_layer = -1, int to_layer = 0, bool bucket = false) {
    if (from_layer < 0) from_layer = h->L - 1;
    for (int l = from_layer; l >= to_layer; --l) {
        const void* in = l == 0 ? x : h->act[l - 1];
        // dW_l = in^T d, db_l = column-sum d, dZ_{l-1} = (d W_l^T) * [z_{l-1} >= 0]  — one launch per layer
        STEP_CALL(h, tnn_dense_bwd(rows, h->w[l], h->w[l + 1], in, h->dact[l], at(h->params, h->w_off[l], h->esz),
                                   at(h->grads, h->w_off[l], h->esz), at(h->grads, h->b_off[l], h->esz),
                                   l > 0 ? h->dact[l - 1] : nullptr, l > 0 ? h->act[l - 1] : nullptr, h->dtype));
        if (bucket) MLP_TRY(allreduce_layer_bucket(h, l));
    }
    return 0;
}

// ---------------------------------------------------------------- bf16 mode
inline void* at16(char* base, int64_t elem_off) { return base + elem_off * 2; }

int mlp16_sync(Mlp* h) {      // refresh the bf16 working copies from the fp32 master parameters
    MLP_TRY(tnn_cast_bf16(h->params, h->w16, h->n_params, 1));
    for (int l = 0; l < h->L; ++l)
        MLP_TRY(tnn_transpose_bf16(at16(h->w16, h->w_off[l]), h->wT16[l], h->w[l], h->w[l + 1]));
    return 0;
}

// emit_t: the hidden layers' activations also leave TRANSPOSED (actT16[l] [w[l+1], rows], the K-contiguous operand of layer
// l + 1's dW product) from the epilogue that produces them — no transpose launches in the backward pass
int mlp16_forward(Mlp* h, const void* x16, int64_t rows, bool emit_t = false) {
    const void* in = x16;
    for (int l = 0; l < h->L; ++l) {
        const bool hidden = l < h->L - 1;
        // z_l = a_{l-1} W_l + b_l : A = a [rows, in] (K = in), B = W_l^T [out, in]
        if (emit_t && hidden)
            STEP_CALL(h, tnn_gemm_bf16_nt_t(rows, h->w[l + 1], h->w[l], in, h->w[l], h->wT16[l], h->w[l], h->act[l], h->w[l + 1],
                                       at(h->params, h->b_off[l], 4), TNN_ACT_RELU, 1, nullptr, 0, h->actT16[l], rows));
        else
            STEP_CALL(h, tnn_gemm_bf16_nt(rows, h->w[l + 1], h->w[l], in, h->w[l], h->wT16[l], h->w[l], h->act[l],
                                     h->w[l + 1], TNN_BF16, at(h->params, h->b_off[l], 4),
                                     hidden ? TNN_ACT_RELU : TNN_ACT_NONE, hidden ? 1 : 0, nullptr, 0));
        in = h->act[l];
    }
    return 0;
}

// The single-GPU bf16 step with the weight gradients consumed where they are produced (tnn_mlp_keep_grads(h, 0)): which form?
//   0  the 25-launch sequence (mlp16_forward + mlp16_backward: separate transposes, bias, loss and partial-sum launches) — shapes
//      the prep launch does not take, or TNN_E_STEP=long
//   1  17 launches (default): see mlp16_step_fused
//   2  13 launches (TNN_E_STEP=ct): a^T / dz^T from the producing GEMMs' epilogues instead of transpose launches — built first,
//      kept for tools/probes/e_step_ab.py, which alternates the forms inside one process (the variable is read at every step)
int mlp16_step_form(const Mlp* h, int64_t rows) {
    if (!(h->bf16 && h->opt_kind == 1 && h->loss_kind == 1 && rows % 64 == 0 && h->w[0] % 64 == 0 && h->w[h->L] % 64 == 0 &&
          h->prep_ws != nullptr && h->L <= 16))
        return 0;
    const char* form = getenv("TNN_E_STEP");
    if (form != nullptr && form[0] == 'l') return 0;
    return form != nullptr && form[0] == 'c' ? 2 : 1;
}

// Single-GPU bf16 step in 4L + 1 = 17 launches for the four 8192-wide layers of configs[4] (25 before):
//   L   forward GEMMs
//   1   prep: loss, dz_L, dz_L^T, Adam's beta powers (tnn_mse_bf16_prep; the loss + partial-sum + transpose launches it replaces)
//   per layer, last to first:
//     1   BOTH K-contiguous operands of dW_l in one launch, just in front of their use: a_{l-1}^T (x^T for the first layer) and
//         dz_l^T (tnn_transpose2_bf16; the last layer's dz^T comes from prep)
//     1   dz_{l-1} = (dz_l W_l^T) * mask (l > 0) — BEFORE dW_l, it reads the bf16 W_l that dW_l's epilogue rewrites
//     1   dW_l = a_{l-1}^T dz_l with Adam on W_l in the epilogue (tnn_gemm_bf16_nt_adam)
//   1   every layer's bias: db_l + Adam on b_l (tnn_bias_bf16_adam_multi; 5.8 us against 16.2 for four launches)
// Why the transposes stay launches (form 2 writes a^T / dz^T from the epilogues of the GEMMs that produce a / dz: + 1.5 us per
// GEMM against 5.9 us per transpose launch, 13 launches): step against step, alternating segments in one process on five boxes,
// form 2 is 0 / 6 / 57 / 84 / 88 us SLOWER than the 25-launch sequence, this form 13 - 22 us FASTER
// (profiles/r05_e_step_ab.txt, r05_e_step_ab_kernels.txt, r05_e_kernels_ab.txt).  The kernel trace says where: the HBM-bound
// dW + Adam launch (1.9 GB, 70 % of the step) takes 25-29 us longer when its operands were written long before (by the forward
// pass; by the dX launch in front of the PREVIOUS dW) than when a transpose launch wrote them just in front of it — those
// launches leave 16 MB of operands in the memory-side cache exactly when a consumer at the HBM roofline is about to re-read them
// from eight XCDs: they are a prefetch.  Also measured and not kept: every dX before the first dW (dW directly behind dW: + 25 us
// each), a bias launch behind each dW, the split-K partners of a tile on one XCD, the bias as a role of the dW launch's
// tile-row-0 workgroups (+ 7.6 us on its critical path — 8 exactly equal rounds of tiles — against 6.8 us for a launch).
// Same arithmetic, element for element, in every form (transposes are exact, the bias sums keep their order).
int mlp16_step_fused(Mlp* h, const void* x16, const void* y16, int64_t rows, void* loss_out, bool ct) {
    const int L = h->L;
    auto f32 = [](void* base, int64_t off) { return (void*)((float*)base + off); };
    MLP_TRY(mlp16_forward(h, x16, rows, ct));
    STEP_CALL(h, tnn_mse_bf16_prep(h->act[L - 1], y16, rows, h->w[L], rows, at(h->grads, h->n_params, 4), loss_out, h->dact[L - 1],
                                   h->dactT16[L - 1], ct ? x16 : nullptr, h->w[0], ct ? h->xT16 : nullptr, h->prep_ws, h->ticket,
                                   h->pows, h->b1, h->b2));
    for (int l = L - 1; l >= 0; --l) {
        const void* in = l == 0 ? x16 : h->act[l - 1];
        void* inT = l == 0 ? h->xT16 : h->actT16[l - 1];
        const int64_t wo = h->w_off[l];
        if (!ct) {
            if (l < L - 1) STEP_CALL(h, tnn_transpose2_bf16(in, inT, rows, h->w[l], h->dact[l], h->dactT16[l], rows, h->w[l + 1]));
            else STEP_CALL(h, tnn_transpose_bf16(in, inT, rows, h->w[l]));
        }
        if (l > 0) {
            if (ct)
                STEP_CALL(h, tnn_gemm_bf16_nt_t(rows, h->w[l], h->w[l + 1], h->dact[l], h->w[l + 1], at16(h->w16, wo), h->w[l + 1],
                                                h->dact[l - 1], h->w[l], nullptr, TNN_ACT_NONE, 0, h->act[l - 1], h->w[l],
                                                h->dactT16[l - 1], rows));
            else
                STEP_CALL(h, tnn_gemm_bf16_nt(rows, h->w[l], h->w[l + 1], h->dact[l], h->w[l + 1], at16(h->w16, wo), h->w[l + 1],
                                              h->dact[l - 1], h->w[l], TNN_BF16, nullptr, TNN_ACT_NONE, 0, h->act[l - 1], h->w[l]));
        }
        // (W_0's [in, out] bf16 copy has no reader inside a step — dX stops at the input —: not written, 2 of 28 B/param of
        // that layer; tnn_mlp_bf16_weights re-derives it from the masters on request)
        STEP_CALL(h, tnn_gemm_bf16_nt_adam(h->w[l], h->w[l + 1], rows, inT, rows, h->dactT16[l], rows, nullptr, f32(h->params, wo),
                                           f32(h->m, wo), f32(h->v, wo), l == 0 ? nullptr : at16(h->w16, wo), h->wT16[l], h->lr,
                                           h->b1, h->b2, h->eps, h->pows));
    }
    const void* dz[16];
    int64_t cols[16];
    void *db[16], *bp[16], *bm[16], *bv[16], *bw[16];
    for (int l = 0; l < L; ++l) {
        const int64_t bo = h->b_off[l];
        dz[l] = h->dact[l]; cols[l] = h->w[l + 1];
        db[l] = f32(h->grads, bo); bp[l] = f32(h->params, bo); bm[l] = f32(h->m, bo); bv[l] = f32(h->v, bo); bw[l] = at16(h->w16, bo);
    }
    STEP_CALL(h, tnn_bias_bf16_adam_multi(L, dz, rows, cols, db, bp, bm, bv, bw, h->lr, h->b1, h->b2, h->eps, h->pows));
    h->w16_first_stale = true;
    return 0;
}

// fused_adam: single-GPU step — Adam consumes dW_l in the GEMM's epilogue (tnn_gemm_bf16_nt_adam; the beta powers were
// advanced by the caller).  dz_{l-1} is then computed BEFORE dW_l, because it reads the bf16 W_l that the epilogue rewrites.
int mlp16_backward(Mlp* h, const void* x16, const void* y16, int64_t rows, int64_t m_global, void* loss_out,
                   bool bucket = false, bool fused_adam = false, bool tick = false) {
    const int L = h->L;
    void* loss_slot = at(h->grads, h->n_params, 4);
    if (h->loss_kind != 1) {
        tnn::set_error("bf16 trainer: only the sum-of-squares loss is implemented");
        return 2;
    }
    // tick: one thread of the loss launch advances Adam's beta powers, its reduction files the loss to loss_out too
    MLP_TRY(tnn_mse_bf16_tick(h->act[L - 1], y16, rows * h->w[L], m_global, loss_slot, tick ? loss_out : nullptr, h->dact[L - 1],
                              tick ? h->pows : nullptr, h->b1, h->b2));
    if (tick) loss_out = nullptr;
    for (int l = L - 1; l >= 0; --l) {
        const void* in = l == 0 ? x16 : h->act[l - 1];
        void* inT = l == 0 ? h->xT16 : h->actT16[l - 1];
        // K-contiguous operands of dW_l = in^T dz: in^T [w[l], rows] and dz^T [w[l+1], rows]
        MLP_TRY(tnn_transpose_bf16(in, inT, rows, h->w[l]));
        MLP_TRY(tnn_transpose_bf16(h->dact[l], h->dactT16[l], rows, h->w[l + 1]));
        // dz_{l-1} = (dz_l W_l^T) * mask : A = dz_l [rows, out] (K = out), B = W_l [in, out]
        auto dz_prev = [&]() -> int {
            if (l == 0) return 0;
            return tnn_gemm_bf16_nt(rows, h->w[l], h->w[l + 1], h->dact[l], h->w[l + 1], at16(h->w16, h->w_off[l]),
                                    h->w[l + 1], h->dact[l - 1], h->w[l], TNN_BF16, nullptr, TNN_ACT_NONE, 0,
                                    h->act[l - 1], h->w[l]);
        };
        if (fused_adam) {
            const int64_t wo = h->w_off[l], bo = h->b_off[l];
            auto f32 = [](void* base, int64_t off) { return (void*)((float*)base + off); };
            MLP_TRY(dz_prev());
            MLP_TRY(tnn_gemm_bf16_nt_adam(h->w[l], h->w[l + 1], rows, inT, rows, h->dactT16[l], rows,
                                          h->keep_grads ? f32(h->grads, wo) : nullptr, f32(h->params, wo), f32(h->m, wo),
                                          f32(h->v, wo), at16(h->w16, wo), h->wT16[l], h->lr, h->b1, h->b2, h->eps, h->pows));
            // db_l + Adam on b_l in one launch
            MLP_TRY(tnn_bias_bf16_adam(h->dact[l], rows, h->w[l + 1], f32(h->grads, bo), f32(h->params, bo), f32(h->m, bo),
                                       f32(h->v, bo), at16(h->w16, bo), h->lr, h->b1, h->b2, h->eps, h->pows));
            continue;
        }
        MLP_TRY(tnn_gemm_bf16_nt(h->w[l], h->w[l + 1], rows, inT, rows, h->dactT16[l], rows,
                                 at(h->grads, h->w_off[l], 4), h->w[l + 1], TNN_F32, nullptr, TNN_ACT_NONE, 0,
                                 nullptr, 0));
        MLP_TRY(tnn_colsum_bf16(h->dact[l], at(h->grads, h->b_off[l], 4), rows, h->w[l + 1]));
        if (bucket) MLP_TRY(allreduce_layer_bucket(h, l));
        MLP_TRY(dz_prev());
    }
    if (loss_out) MLP_TRY(tnn_memcpy_d2d(loss_out, loss_slot, 4));
    return 0;
}

int mlp16_update(Mlp* h) {
    if (h->opt_kind != 1) {
        tnn::set_error("bf16 trainer: only Adam is implemented");
        return 2;
    }
    // per layer: W (also refreshing W^T for the forward GEMM) then b; the beta powers advance once, up front
    auto f32 = [](void* base, int64_t off) { return (void*)((float*)base + off); };
    for (int l = 0; l < h->L; ++l) {
        const int64_t wo = h->w_off[l], bo = h->b_off[l];
        MLP_TRY(tnn_adam_master_bf16_2d(f32(h->params, wo), f32(h->grads, wo), f32(h->m, wo), f32(h->v, wo),
                                        at16(h->w16, wo), h->wT16[l], h->w[l], h->w[l + 1], h->lr, h->b1, h->b2,
                                        h->eps, h->pows, l == 0));
        MLP_TRY(tnn_adam_master_bf16_2d(f32(h->params, bo), f32(h->grads, bo), f32(h->m, bo), f32(h->v, bo),
                                        at16(h->w16, bo), nullptr, 1, h->w[l + 1], h->lr, h->b1, h->b2, h->eps,
                                        h->pows, 0));
    }
    return 0;
}

// Data-parallel bf16 step with the optimizer SHARDED over the ranks (configs[4]: 268 M parameters, 8 GPUs).  Instead of
// "all-reduce 1.07 GB of fp32 gradients, every rank runs the same 8.6 GB optimizer pass" (run.py:82-83 taken literally):
//   per layer, right behind its dW launch and on the communication stream while the library stream goes on with the next
//   layer's backward:  reduce-scatter of dW_l as bf16 (rank r receives the summed rows [r in/W, (r+1) in/W))  ->  Adam on
//   those rows of the fp32 master weights (core/optimizer.py:67-79; tnn_adam_master_g16)  ->  all-gather of the refreshed
//   bf16 rows into every rank's working copy.
// Bytes on the links per step and rank: 2 (W-1)/W x 2 B per parameter (what ONE bf16 all-reduce moves; the fp32
// all-reduce moves twice that); optimizer traffic 30 B per parameter / W.  Biases and the loss: one small fp32 all-reduce,
// replicated Adam.  Each rank's fp32 master / m / v arenas are authoritative for its own rows only; the bf16 working copy
// (tnn_mlp_bf16_weights) is complete and identical on every rank.  dW is rounded to bf16 once before the sum.
bool zero_step_fits(const Mlp* h, int world) {
    if (!h->bf16 || h->opt_kind != 1 || h->loss_kind != 1) return false;
    for (int l = 0; l < h->L; ++l)
        if (h->w[l] % world || ((h->w[l] / world) * h->w[l + 1]) % 8) return false;
    return true;
}

int mlp16_step_zero(Mlp* h, const void* x16, const void* y16, int64_t rows, int rank, int world, void* loss_out) {
    const int L = h->L;
    if (!h->g16) MLP_TRY(tnn_malloc((size_t)h->arena * 2, (void**)&h->g16));
    if (!h->bias_g) {
        int64_t nb = 0;
        h->bias_g_off.clear();
        for (int l = 0; l < L; ++l) { h->bias_g_off.push_back(nb); nb += h->w[l + 1]; }
        h->bias_g_off.push_back(nb);                      // the loss
        MLP_TRY(tnn_malloc((size_t)(nb + 1) * 4, (void**)&h->bias_g));
    }
    const int64_t nb = h->bias_g_off[L];
    auto f32 = [](void* base, int64_t off) { return (void*)((float*)base + off); };
    MLP_TRY(mlp16_forward(h, x16, rows));
    MLP_TRY(tnn_mse_bf16_tick(h->act[L - 1], y16, rows * h->w[L], rows * world, h->bias_g + nb, nullptr, h->dact[L - 1], h->pows,
                              h->b1, h->b2));
    int rc = 0, chains = 0;
    for (int l = L - 1; l >= 0 && !rc; --l) {
        const void* in = l == 0 ? x16 : h->act[l - 1];
        void* inT = l == 0 ? h->xT16 : h->actT16[l - 1];
        const int64_t wo = h->w_off[l];
        rc = tnn_transpose_bf16(in, inT, rows, h->w[l]);
        if (!rc) rc = tnn_transpose_bf16(h->dact[l], h->dactT16[l], rows, h->w[l + 1]);
        // dz_{l-1} reads the bf16 W_l that this layer's all-gather rewrites: it is enqueued BEFORE the chain opens
        if (!rc && l > 0)
            rc = tnn_gemm_bf16_nt(rows, h->w[l], h->w[l + 1], h->dact[l], h->w[l + 1], at16(h->w16, wo), h->w[l + 1],
                                  h->dact[l - 1], h->w[l], TNN_BF16, nullptr, TNN_ACT_NONE, 0, h->act[l - 1], h->w[l]);
        if (!rc)
            rc = tnn_gemm_bf16_nt(h->w[l], h->w[l + 1], rows, inT, rows, h->dactT16[l], rows, at16(h->g16, wo), h->w[l + 1],
                                  TNN_BF16, nullptr, TNN_ACT_NONE, 0, nullptr, 0);
        if (!rc) rc = tnn_bias_bf16_adam(h->dact[l], rows, h->w[l + 1], h->bias_g + h->bias_g_off[l], nullptr, nullptr, nullptr,
                                         nullptr, h->lr, h->b1, h->b2, h->eps, nullptr);
        if (rc) break;
        const int64_t n_shard = h->w[l] / world * h->w[l + 1], so = wo + (int64_t)rank * n_shard;
        rc = tnn_comm_chain_begin();
        if (rc) break;
        rc = tnn_reduce_scatter(at16(h->g16, wo), at16(h->g16, so), n_shard, TNN_BF16);
        if (!rc)
            rc = tnn_adam_master_g16(f32(h->params, so), at16(h->g16, so), f32(h->m, so), f32(h->v, so), at16(h->w16, so),
                                     n_shard, h->lr, h->b1, h->b2, h->eps, h->pows);
        if (!rc) rc = tnn_allgather(at16(h->w16, so), at16(h->w16, wo), n_shard, TNN_BF16);
        const int rc_end = tnn_comm_chain_end();
        if (!rc) rc = rc_end;
        ++chains;
    }
    // bias gradients + loss: one small all-reduce on the library stream, the same Adam on every rank
    if (!rc) rc = tnn_allreduce(h->bias_g, nb + 1, TNN_F32, TNN_RSUM);
    for (int l = 0; l < L && !rc; ++l) {
        const int64_t bo = h->b_off[l];
        rc = tnn_adam_master_bf16_2d(f32(h->params, bo), h->bias_g + h->bias_g_off[l], f32(h->m, bo), f32(h->v, bo),
                                     at16(h->w16, bo), nullptr, 1, h->w[l + 1], h->lr, h->b1, h->b2, h->eps, h->pows, 0);
    }
    // W_l^T for the next forward, layer by layer as the chains land (issue order = last layer first)
    for (int l = L - 1; l >= 0 && !rc && chains > 0; --l, --chains) {
        rc = tnn_comm_wait_oldest();
        if (!rc) rc = tnn_transpose_bf16(at16(h->w16, h->w_off[l]), h->wT16[l], h->w[l], h->w[l + 1]);
    }
    if (rc) { (void)tnn_comm_join(); return rc; }
    MLP_TRY(tnn_memcpy_d2d(at(h->grads, h->n_params, 4), h->bias_g + nb, 4));
    if (loss_out) MLP_TRY(tnn_memcpy_d2d(loss_out, h->bias_g + nb, 4));
    h->masters_world = world > 1 ? world : 0;
    return 0;
}

// Collective: all-gather every rank's OWNED fp32 rows of W_l (master weights and both Adam moments) so that the three arenas
// are whole on every rank again — what a checkpoint (MLPTrainer.state_dict) or any reader of the fp32 parameters needs after
// sharded-optimizer steps.  Biases are replicated already.  3 x 4 B per parameter over the links: checkpoint-time only.
int mlp16_gather_masters(Mlp* h) {
    if (h->masters_world <= 1) return 0;
    int rank = 0, world = 1;
    MLP_TRY(tnn_comm_world(&rank, &world));
    if (world != h->masters_world) {
        tnn::set_error("tnn_mlp_gather_masters: the optimizer was sharded over %d ranks, the communicator now has %d",
                       h->masters_world, world);
        return 2;
    }
    char* arenas[3] = {h->params, h->m, h->v};
    for (int l = 0; l < h->L; ++l) {
        const int64_t n_shard = h->w[l] / world * h->w[l + 1], wo = h->w_off[l], so = wo + (int64_t)rank * n_shard;
        for (char* a : arenas) MLP_TRY(tnn_allgather(at(a, so, 4), at(a, wo, 4), n_shard, TNN_F32));
    }
    h->masters_world = 0;
    return 0;
}

// The merged head + hidden-backward launch with the softmax statistics taken from memory (tnn_mlp_head_bwd_tick_ext):
// nothing couples the rows inside that launch, so it walks them in blocks of 128 — the 2L - 2 launch step for batches of more
// than 128 rows.  One GPU (129 .. 1024 rows): the hidden layer's forward in its row-panel form (tnn_dense_fwd_rows_head_stats:
// a workgroup owns 16 whole rows and finishes their logits and statistics itself).  Data parallel (129 .. 512 rows per rank,
// config D at 2 / 4 ranks): the tiled forward whose tail reduces the statistics per 128-row block behind arrival counters
// (tnn_dense_fwd_head_partials_stats) — the exchange needs ONE pair per rank.  Measured, same box, us per step at 256 / 512 /
// 1024 rows: row-panel forward 26.9 / 34.7 / 45.5, counter tail 29.6 / 38.6 / 54.8, the 7-launch form below 35.4 / 41.4 / 52.7
// (profiles/r03_rows_sweep.txt; the switches that selected the forms for that sweep are gone).
bool head_fits_row_blocks(const Mlp* h, int64_t rows, bool sharded) {
    const int L = h->L;
    const int64_t row_blocks_max = 1024;             // 8 blocks of 128: the forward tail's counters and pair slots (h->ticket)
    (void)sharded;
    // the tuned 128 -> 10 head, or any head the generic merged kernel takes (tnn_mlp_head_bwd_fits: hidden width a multiple of 16
    // up to 256, <= 16 classes — the reference's own 30 -> 10, padded to 32)
    const bool head_ok = h->w[L - 1] % 16 == 0 && h->w[L - 1] >= 16 && h->w[L - 1] <= 256 && h->w[L] >= 1 && h->w[L] <= 16;
    return h->dtype == TNN_F32 && !h->bf16 && h->opt_kind == 1 && h->loss_kind == 0 && L >= 3 && head_ok &&
           h->w[L - 2] % 16 == 0 && rows > 128 && rows <= row_blocks_max && rows <= 1024 && h->zpart != nullptr;
}

// the data-parallel step's merged form (statistics from memory at every row count): the same heads, 1 .. 1024 rows per rank
bool head_fits_sharded(const Mlp* h, int64_t rows) {
    return rows <= 128 ? head_fits_row_blocks(h, rows + 128, true) : head_fits_row_blocks(h, rows, true);
}

// limits of the single-workgroup loss kernel (tnn_softmax_nll_fused_tick)
bool head_fits_one_workgroup(const Mlp* h, int64_t rows) {
    // classifier heads (<= 16 classes): the one-thread-per-row kernel, up to 1024 rows; wider heads: the LDS image of the
    // element-parallel kernel (tnn_softmax_nll_fused_tick)
    return !h->bf16 && h->loss_kind == 0 && rows <= 1024 &&
           (h->w[h->L] <= 16 || rows * h->w[h->L] <= (h->dtype == TNN_F32 ? 4096 : 2048));
}

// Adam on ONE layer's parameters (W_l and b_l are contiguous in the arenas).  advance: first call of the step
// (advances the beta powers); loss_out: also file the (already reduced) loss.
int adam_layer(Mlp* h, int l, bool advance, void* loss_out) {
    const int64_t wo = h->w_off[l], bo = h->b_off[l];
    void* loss_slot = at(h->grads, h->n_params, h->esz);
    if (h->bf16) {
        auto f32 = [](void* base, int64_t off) { return (void*)((float*)base + off); };
        MLP_TRY(tnn_adam_master_bf16_2d(f32(h->params, wo), f32(h->grads, wo), f32(h->m, wo), f32(h->v, wo),
                                        at16(h->w16, wo), h->wT16[l], h->w[l], h->w[l + 1], h->lr, h->b1, h->b2,
                                        h->eps, h->pows, advance ? 1 : 0));
        MLP_TRY(tnn_adam_master_bf16_2d(f32(h->params, bo), f32(h->grads, bo), f32(h->m, bo), f32(h->v, bo),
                                        at16(h->w16, bo), nullptr, 1, h->w[l + 1], h->lr, h->b1, h->b2, h->eps,
                                        h->pows, 0));
        if (loss_out) MLP_TRY(tnn_memcpy_d2d(loss_out, loss_slot, 4));
        return 0;
    }
    const int64_t count = h->w[l] * h->w[l + 1] + h->w[l + 1];
    return tnn_adam_ex(at(h->params, wo, h->esz), at(h->grads, wo, h->esz), at(h->m, wo, h->esz), at(h->v, wo, h->esz),
                       count, h->lr, h->b1, h->b2, h->eps, h->pows, nullptr, h->dtype, advance ? 1 : 0,
                       loss_out ? loss_slot : nullptr, loss_out);
}

int check_rows(Mlp* h, int64_t rows, const char* fn) {
    if (!h) { tnn::set_error("%s: NULL handle", fn); return 2; }
    if (rows <= 0 || rows > h->max_rows) {
        tnn::set_error("%s: rows %lld outside (0, %lld]", fn, (long long)rows, (long long)h->max_rows);
        return 2;
    }
    return 0;
}

}  // namespace

extern "C" {

int tnn_mlp_create(int n_layers, const int64_t* widths, int64_t max_rows, int loss_kind,
                   int opt_kind, double lr, double b1, double b2, double eps, int dtype,
                   void** handle) {
    if (n_layers < 1 || !widths || max_rows < 1 || !handle) {
        tnn::set_error("tnn_mlp_create: bad arguments");
        return 2;
    }
    if (dtype != TNN_F32 && dtype != TNN_F64 && dtype != TNN_BF16) {
        tnn::set_error("tnn_mlp_create: dtype %d is not a float type", dtype);
        return 2;
    }
    if (loss_kind < 0 || loss_kind > 1 || opt_kind < 0 || opt_kind > 5) {
        tnn::set_error("tnn_mlp_create: loss_kind %d / opt_kind %d unknown", loss_kind, opt_kind);
        return 2;
    }
    Mlp* h = new Mlp();
    h->L = n_layers;
    h->w.assign(widths, widths + n_layers + 1);
    h->max_rows = max_rows;
    h->loss_kind = loss_kind;
    h->opt_kind = opt_kind;
    h->dtype = dtype;
    h->lr = lr; h->b1 = b1; h->b2 = b2; h->eps = eps;
    h->bf16 = dtype == TNN_BF16;
    h->esz = dtype == TNN_F64 ? 8 : 4;            // element size of the four arenas (fp32 in bf16 mode)
    int64_t off = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (widths[l] < 1 || widths[l + 1] < 1) {
            delete h;
            tnn::set_error("tnn_mlp_create: layer widths must be positive");
            return 2;
        }
        h->w_off.push_back(off);
        off += widths[l] * widths[l + 1];
        h->b_off.push_back(off);
        off += widths[l + 1];
    }
    h->n_params = off;
    h->arena = (off + 1 + 3) / 4 * 4;   // + loss slot, padded to 16 B
    size_t bytes = (size_t)h->arena * h->esz;
    int rc = 0;
    rc |= tnn_malloc(bytes, (void**)&h->params);
    rc |= tnn_malloc(bytes, (void**)&h->grads);
    rc |= tnn_malloc(bytes, (void**)&h->m);
    rc |= tnn_malloc(bytes, (void**)&h->v);
    rc |= tnn_malloc(4 * sizeof(double), &h->pows);
    rc |= tnn_malloc(64 * 2 * 8 + 64, &h->stats);   // {max, sum-exp}: one pair, or one per 16-row panel (<= 64) of the row-panel forward (behind a merged pair: + 16 B)
    rc |= tnn_malloc(256, &h->ticket);          // 16 arrival counters + the row blocks' {max, sum-exp} pairs (tnn_dense_fwd_head_partials_stats)
    if (!rc) rc |= tnn_memset(h->ticket, 0, 256);
    if (n_layers >= 2 && dtype == TNN_F32)
        rc |= tnn_malloc((size_t)((widths[n_layers - 1] + 15) / 16 * max_rows * widths[n_layers]) * 4, &h->zpart);
    if (n_layers >= 3 && dtype == TNN_F32 && loss_kind == 0 && max_rows > 128 && !rc)     // (the merged head's row-block hand-off memory)
        rc |= tnn_mlp_head_bwd_reserve(max_rows, widths[n_layers - 2], widths[n_layers - 1], widths[n_layers]);
    const size_t act_esz = h->bf16 ? 2 : h->esz;
    for (int l = 0; l < n_layers && !rc; ++l) {
        void *a = nullptr, *d = nullptr;
        rc |= tnn_malloc((size_t)(max_rows * widths[l + 1]) * act_esz, &a);
        rc |= tnn_malloc((size_t)(max_rows * widths[l + 1]) * act_esz, &d);
        h->act.push_back(a);
        h->dact.push_back(d);
        if (h->bf16) {
            void *wt = nullptr, *at_ = nullptr, *dt = nullptr;
            rc |= tnn_malloc((size_t)(widths[l] * widths[l + 1]) * 2, &wt);
            rc |= tnn_malloc((size_t)(max_rows * widths[l + 1]) * 2, &at_);
            rc |= tnn_malloc((size_t)(max_rows * widths[l + 1]) * 2, &dt);
            h->wT16.push_back(wt);
            h->actT16.push_back(at_);
            h->dactT16.push_back(dt);
        }
    }
    if (h->bf16 && !rc) {
        rc |= tnn_malloc((size_t)h->arena * 2, (void**)&h->w16);
        rc |= tnn_malloc((size_t)(max_rows * widths[0]) * 2, &h->xT16);
        rc |= tnn_malloc((size_t)((max_rows + 63) / 64 * ((widths[n_layers] + 63) / 64)) * sizeof(double), &h->prep_ws);
        // hand-off memory of the skinny GEMMs, allocated now: the first step may already be inside a hipGraph capture
        for (int l = 0; l < n_layers && !rc; ++l) {
            rc |= tnn_gemm_bf16_reserve(max_rows, widths[l + 1], widths[l]);
            rc |= tnn_gemm_bf16_reserve(max_rows, widths[l], widths[l + 1]);
        }
    }
    if (!rc) {
        rc |= tnn_memset(h->params, 0, bytes);
        rc |= tnn_memset(h->grads, 0, bytes);
        rc |= tnn_memset(h->m, 0, bytes);
        rc |= tnn_memset(h->v, 0, bytes);
        const double init[4] = {1.0, 1.0, 0.0, 0.0};
        rc |= tnn_memcpy_h2d(h->pows, init, sizeof(init));
    }
    if (rc) {
        tnn_mlp_destroy(h);
        return 1;
    }
    *handle = h;
    return 0;
}

int tnn_mlp_destroy(void* handle) {
    Mlp* h = (Mlp*)handle;
    if (!h) return 0;
    tnn_free(h->params); tnn_free(h->grads); tnn_free(h->m); tnn_free(h->v);
    tnn_free(h->pows); tnn_free(h->stats); tnn_free(h->ticket); tnn_free(h->stats_all); tnn_free(h->zpart);
    for (void* p : h->act) tnn_free(p);
    for (void* p : h->dact) tnn_free(p);
    for (void* p : h->wT16) tnn_free(p);
    for (void* p : h->actT16) tnn_free(p);
    for (void* p : h->dactT16) tnn_free(p);
    tnn_free(h->w16); tnn_free(h->xT16); tnn_free(h->g16); tnn_free(h->bias_g); tnn_free(h->prep_ws);
    delete h;
    return 0;
}

int tnn_mlp_arena(void* handle, void** params, void** grads, void** m, void** v, int64_t* n_params) {
    Mlp* h = (Mlp*)handle;
    if (!h) { tnn::set_error("tnn_mlp_arena: NULL handle"); return 2; }
    if (params) *params = h->params;
    if (grads) *grads = h->grads;
    if (m) *m = h->m;
    if (v) *v = h->v;
    if (n_params) *n_params = h->n_params;
    return 0;
}

int tnn_mlp_optimizer_state(void* handle, void** pows_f64) {
    Mlp* h = (Mlp*)handle;
    if (!h || !pows_f64) { tnn::set_error("tnn_mlp_optimizer_state: bad arguments"); return 2; }
    *pows_f64 = h->pows;
    return 0;
}

int tnn_mlp_param_offset(void* handle, int layer, int which, int64_t* offset, int64_t* count) {
    Mlp* h = (Mlp*)handle;
    if (!h || layer < 0 || layer >= h->L || which < 0 || which > 1) {
        tnn::set_error("tnn_mlp_param_offset: bad arguments");
        return 2;
    }
    if (offset) *offset = which == 0 ? h->w_off[layer] : h->b_off[layer];
    if (count) *count = which == 0 ? h->w[layer] * h->w[layer + 1] : h->w[layer + 1];
    return 0;
}

int tnn_mlp_forward(void* handle, const void* x, int64_t rows, void* logits) {
    Mlp* h = (Mlp*)handle;
    MLP_TRY(check_rows(h, rows, "tnn_mlp_forward"));
    h->call_idx = 0;
    if (h->bf16) {
        MLP_TRY(mlp16_forward(h, x, rows));
        if (logits) MLP_TRY(tnn_memcpy_d2d(logits, h->act[h->L - 1], (size_t)(rows * h->w[h->L]) * 2));
        return 0;
    }
    MLP_TRY(mlp_forward(h, x, rows));
    if (logits)
        MLP_TRY(tnn_memcpy_d2d(logits, h->act[h->L - 1], (size_t)(rows * h->w[h->L]) * h->esz));
    return 0;
}

static int mlp_forward_stats(Mlp* h, const void* x, int64_t rows, void* stats);

int tnn_mlp_forward_stats(void* handle, const void* x, int64_t rows, void* stats) {
    Mlp* h = (Mlp*)handle;
    MLP_TRY(check_rows(h, rows, "tnn_mlp_forward_stats"));
    h->call_idx = 0;
    return mlp_forward_stats(h, x, rows, stats);
}

static int mlp_forward_stats(Mlp* h, const void* x, int64_t rows, void* stats) {
    if (h->bf16) return mlp16_forward(h, x, rows);
    MLP_TRY(mlp_forward(h, x, rows));
    if (h->loss_kind == 0)
        STEP_CALL(h, tnn_softmax_nll_stats(h->act[h->L - 1], rows, h->w[h->L], stats ? stats : h->stats,
                                           h->dtype));
    return 0;
}

static int mlp_backward_impl(Mlp* h, const void* x, const void* y, int64_t rows, int64_t m_global,
                             const void* stats, void* loss_out, bool bucket) {
    MLP_TRY(check_rows(h, rows, "tnn_mlp_backward"));
    if (h->bf16) return mlp16_backward(h, x, y, rows, m_global, loss_out, bucket);
    const int L = h->L;
    void* loss_slot = at(h->grads, h->n_params, h->esz);
    if (h->loss_kind == 0)
        STEP_CALL(h, tnn_softmax_nll_fwd_bwd(h->act[L - 1], y, rows, h->w[L], m_global,
                                             stats ? stats : h->stats, loss_slot, h->dact[L - 1], h->dtype));
    else
        STEP_CALL(h, tnn_mse_fwd_bwd(h->act[L - 1], y, rows * h->w[L], m_global, loss_slot, h->dact[L - 1],
                                     h->dtype));
    MLP_TRY(mlp_backward_layers(h, x, rows, -1, 0, bucket));
    if (loss_out) MLP_TRY(tnn_memcpy_d2d(loss_out, loss_slot, h->esz));
    return 0;
}

int tnn_mlp_backward(void* handle, const void* x, const void* y, int64_t rows, int64_t m_global,
                     const void* stats, void* loss_out) {
    if (handle) ((Mlp*)handle)->call_idx = 0;
    return mlp_backward_impl((Mlp*)handle, x, y, rows, m_global, stats, loss_out, false);
}

static int mlp_update(Mlp* h);

int tnn_mlp_update(void* handle) {
    Mlp* h = (Mlp*)handle;
    if (!h) { tnn::set_error("tnn_mlp_update: NULL handle"); return 2; }
    h->call_idx = 0;
    return mlp_update(h);
}

static int mlp_update(Mlp* h) {
    if (h->bf16) return mlp16_update(h);
    if (h->opt_kind == 0)
        STEP_CALL(h, tnn_sgd(h->params, h->grads, h->n_params, h->lr, h->dtype));
    else if (h->opt_kind >= 2)   // Momentum / RMSProp / Adagrad / Adadelta: m, v are the two state vectors; b1, b2 = a, b
        STEP_CALL(h, tnn_optim_step(h->opt_kind - 2, h->params, h->grads, h->m, h->v, nullptr, h->n_params, h->lr,
                                    h->b1, h->b2, h->eps, h->dtype));
    else
        STEP_CALL(h, tnn_adam(h->params, h->grads, h->m, h->v, h->n_params, h->lr, h->b1, h->b2, h->eps,
                              h->pows, nullptr, h->dtype));
    return 0;
}

int tnn_mlp_step(void* handle, const void* x, const void* y, int64_t rows, void* loss_out) {
    Mlp* h = (Mlp*)handle;
    MLP_TRY(check_rows(h, rows, "tnn_mlp_step"));
    h->call_idx = 0;
    if (h->bf16 && h->opt_kind == 1 && h->loss_kind == 1 && !h->keep_grads) {
        // bf16 trainer, single GPU, weight gradients not wanted in the arena (tnn_mlp_keep_grads(h, 0)): forward | beta
        // powers | backward with Adam in the dW epilogues — no optimizer launch over the weights, no weight-gradient
        // round trip through HBM (8 of the 36 bytes per parameter and step; measured 8192 x 8192 x 512: 385-405 us against
        // 450-495 for GEMM + optimizer).  With the gradient ALSO stored the fused launch is slower than the two (517 us):
        // keep_grads stays on the separate launches.
        if (const int form = mlp16_step_form(h, rows)) return mlp16_step_fused(h, x, y, rows, loss_out, form == 2);
        MLP_TRY(mlp16_forward(h, x, rows));
        return mlp16_backward(h, x, y, rows, rows, loss_out, false, true, true);
    }
    if (!h->bf16 && h->dtype == TNN_F32 && h->opt_kind == 1 && !h->keep_grads && h->n_params >= (1 << 22)) {
        // large fp32 nets, single GPU, weight gradients not wanted in the arena: Adam in the epilogue of every dW GEMM
        // (tnn_gemm_tn_adam).  The fp32 products are MFMA-bound, so the optimizer's 24 B per parameter ride under them and
        // the separate pass over the arena disappears.  dz_{l-1} is computed BEFORE dW_l: it reads the W_l the epilogue
        // rewrites.
        // Launches per step: L forward | loss (2; Adam's beta powers advanced by a thread of it, the loss filed to loss_out by
        // its reduction) | per layer: dz_{l-1} (l > 0) and ONE launch for dW_l + Adam on W_l + db_l + Adam on b_l
        // (tnn_gemm_tn_adam_bias) — 7 for the two-layer 4096-wide net of configs[2] (15 with the column reductions, bias
        // optimizer launches, beta-power tick and loss copy as launches of their own).
        const int L = h->L;
        void* loss_slot = at(h->grads, h->n_params, h->esz);
        MLP_TRY(mlp_forward_stats(h, x, rows, nullptr));
        if (h->loss_kind == 0) {
            MLP_TRY(tnn_softmax_nll_fwd_bwd(h->act[L - 1], y, rows, h->w[L], rows, h->stats, loss_slot, h->dact[L - 1], h->dtype));
            MLP_TRY(tnn_adam_tick(h->pows, h->b1, h->b2));
            if (loss_out) MLP_TRY(tnn_memcpy_d2d(loss_out, loss_slot, h->esz));
        } else {
            MLP_TRY(tnn_mse_fwd_bwd_tick(h->act[L - 1], y, rows * h->w[L], rows, loss_slot, loss_out, h->dact[L - 1], h->dtype,
                                         h->pows, h->b1, h->b2));
        }
        for (int l = L - 1; l >= 0; --l) {
            const void* in = l == 0 ? x : h->act[l - 1];
            const int64_t wo = h->w_off[l], bo = h->b_off[l];
            if (l > 0)
                MLP_TRY(tnn_gemm_mask(0, 1, rows, h->w[l], h->w[l + 1], h->dact[l], h->w[l + 1], at(h->params, wo, 4),
                                      h->w[l + 1], h->act[l - 1], h->w[l], h->dact[l - 1], h->w[l], h->dtype));
            MLP_TRY(tnn_gemm_tn_adam_bias(h->w[l], h->w[l + 1], rows, in, h->w[l], h->dact[l], h->w[l + 1], nullptr,
                                          at(h->params, wo, 4), at(h->m, wo, 4), at(h->v, wo, 4), at(h->grads, bo, 4),
                                          at(h->params, bo, 4), at(h->m, bo, 4), at(h->v, bo, 4), h->lr, h->b1, h->b2, h->eps,
                                          h->pows, h->dtype));
        }
        return 0;
    }
    if (h->loss_kind != 0) {
        MLP_TRY(mlp_forward_stats(h, x, rows, nullptr));
        MLP_TRY(mlp_backward_impl(h, x, y, rows, rows, nullptr, loss_out, false));
        return mlp_update(h);
    }
    // unsharded softmax head: stats + loss + dz in one launch; the loss goes straight to loss_out
    // (e.g. one slot of a per-step loss history) or, by default, to the slot behind the gradient arena
    const int L = h->L;
    void* loss_dst = loss_out ? loss_out : at(h->grads, h->n_params, h->esz);
    int head_multi = 0, head_bwd = 0;
    if (!h->bf16 && h->opt_kind == 1 && L >= 2)
        MLP_TRY(tnn_mlp_head_fits(rows, h->w[L - 1], h->w[L], h->dtype, &head_multi));
    // any other head the merged head + hidden-backward launch takes (generic kernel: hidden widths multiples of 16 up to 256,
    // <= 16 classes): the same 2L - 2 launch step
    if (!head_multi && !h->bf16 && h->opt_kind == 1 && L >= 3 && h->zpart != nullptr)
        MLP_TRY(tnn_mlp_head_bwd_fits(rows, h->w[L - 2], h->w[L - 1], h->w[L], h->dtype, &head_bwd));
    if (head_multi || head_bwd) {
        // 2L - 1 launches (5 for the MNIST net; 2L - 2 = 4 with the merge below): forward of the hidden layers | the classifier head as ONE multi-workgroup
        // launch (last Dense forward + loss + last Dense backward + Adam's beta powers) | backward of the hidden layers,
        // the first layer's carrying the whole optimizer
        // the hidden layer in front of the classifier also emits the logits as per-tile partial sums (its activations
        // are in registers there), so no workgroup of the head has to redo a W
        MLP_TRY(mlp_forward(h, x, rows, L - 2));
        STEP_CALL(h, tnn_dense_fwd_head_partials(rows, h->w[L - 1], h->w[L - 2], L > 2 ? h->act[L - 3] : x, h->w[L - 2],
                                                 at(h->params, h->w_off[L - 2], h->esz), h->w[L - 1],
                                                 at(h->params, h->b_off[L - 2], h->esz), TNN_ACT_RELU, 1, h->act[L - 2],
                                                 h->w[L - 1], at(h->params, h->w_off[L - 1], h->esz), h->w[L], h->zpart,
                                                 h->dtype));
        if (head_bwd || (L >= 3 && h->w[L - 2] % 16 == 0)) {
            // 2L - 2 launches (4 for the MNIST net): the head's launch also carries the backward of the hidden layer in
            // front of it — its tiles derive their slice of that layer's dz from the partial logits themselves
            STEP_CALL(h, tnn_mlp_head_bwd_tick(rows, h->w[L - 2], h->w[L - 1], h->w[L], h->act[L - 3],
                                               at(h->params, h->w_off[L - 2], h->esz), h->act[L - 2],
                                               at(h->params, h->w_off[L - 1], h->esz), at(h->params, h->b_off[L - 1], h->esz),
                                               y, h->zpart, h->act[L - 1], h->dact[L - 1], h->stats, loss_dst,
                                               at(h->grads, h->w_off[L - 1], h->esz), at(h->grads, h->b_off[L - 1], h->esz),
                                               at(h->grads, h->w_off[L - 2], h->esz), at(h->grads, h->b_off[L - 2], h->esz),
                                               h->dact[L - 3], h->dtype, h->pows, h->b1, h->b2));
            MLP_TRY(mlp_backward_layers(h, x, rows, L - 3, 1));
        } else {
            STEP_CALL(h, tnn_mlp_head_tick(rows, h->w[L - 1], h->w[L], h->act[L - 2],
                                           at(h->params, h->w_off[L - 1], h->esz), at(h->params, h->b_off[L - 1], h->esz), y,
                                           h->zpart, h->act[L - 1], h->dact[L - 1], h->stats, loss_dst,
                                           at(h->grads, h->w_off[L - 1], h->esz), at(h->grads, h->b_off[L - 1], h->esz),
                                           h->dact[L - 2], h->dtype, h->pows, h->b1, h->b2));
            MLP_TRY(mlp_backward_layers(h, x, rows, L - 2, 1));
        }
        const int64_t rest = h->w_off[1];
        STEP_CALL(h, tnn_dense_bwd_first_adam(rows, h->w[0], h->w[1], x, h->dact[0],
                                              h->keep_grads ? at(h->grads, h->w_off[0], h->esz) : nullptr,
                                              at(h->grads, h->b_off[0], h->esz), at(h->params, h->w_off[0], h->esz),
                                              at(h->m, h->w_off[0], h->esz), at(h->v, h->w_off[0], h->esz),
                                              at(h->params, h->b_off[0], h->esz), at(h->m, h->b_off[0], h->esz),
                                              at(h->v, h->b_off[0], h->esz), at(h->params, rest, h->esz),
                                              at(h->grads, rest, h->esz), at(h->m, rest, h->esz),
                                              at(h->v, rest, h->esz), h->n_params - rest, h->lr, h->b1, h->b2, h->eps,
                                              h->pows, h->dtype));
        return 0;
    }
    if (head_fits_row_blocks(h, rows, false)) {
        // 129 .. 1024 rows, 2L - 2 launches (4 for the MNIST net) like the <= 128-row step: the hidden layer's forward leaves the
        // whole-batch {max, sum-exp} as pairs in memory, the merged head launch reads them
        // and walks the rows in blocks of 128 (dW / db / loss accumulated in registers), the first layer's backward carries
        // the optimizer.  (Before: 7 launches — three forward, a one-workgroup loss, three backward: 35.4 us at 256 rows.)
        MLP_TRY(mlp_forward(h, x, rows, L - 2));
        // one GPU: the hidden layer's forward in its row-panel form — a workgroup owns 16 whole rows, finishes their logits and
        // their softmax statistics itself and leaves one {max, sum-exp} pair per panel (no arrival counter, no re-read of
        // partial logits at the tail of the launch); the merged launch merges the pairs (n_pairs < 0: whole logits)
        const int n_panels = (int)((rows + 15) / 16);
        const bool tuned_head = h->w[L - 1] == 128 && h->w[L] == 10;       // (any other head: the counter tail, one merged pair)
        if (tuned_head && h->w[L - 2] % 4 == 0)
            STEP_CALL(h, tnn_dense_fwd_rows_head_stats(rows, h->w[L - 1], h->w[L - 2], h->act[L - 3], h->w[L - 2],
                                                       at(h->params, h->w_off[L - 2], h->esz), h->w[L - 1],
                                                       at(h->params, h->b_off[L - 2], h->esz), TNN_ACT_RELU, 1, h->act[L - 2],
                                                       h->w[L - 1], at(h->params, h->w_off[L - 1], h->esz), h->w[L], h->zpart,
                                                       at(h->params, h->b_off[L - 1], h->esz), h->stats, h->dtype));
        else
            STEP_CALL(h, tnn_dense_fwd_head_partials_stats(rows, h->w[L - 1], h->w[L - 2], h->act[L - 3], h->w[L - 2],
                                                           at(h->params, h->w_off[L - 2], h->esz), h->w[L - 1],
                                                           at(h->params, h->b_off[L - 2], h->esz), TNN_ACT_RELU, 1, h->act[L - 2],
                                                           h->w[L - 1], at(h->params, h->w_off[L - 1], h->esz), h->w[L], h->zpart,
                                                           at(h->params, h->b_off[L - 1], h->esz), y, h->ticket, h->stats, 0,
                                                           h->dtype));
        STEP_CALL(h, tnn_mlp_head_bwd_tick_ext(rows, rows, h->w[L - 2], h->w[L - 1], h->w[L], h->act[L - 3],
                                               at(h->params, h->w_off[L - 2], h->esz), h->act[L - 2],
                                               at(h->params, h->w_off[L - 1], h->esz), at(h->params, h->b_off[L - 1], h->esz),
                                               y, h->zpart, h->stats, (tuned_head && h->w[L - 2] % 4 == 0) ? -n_panels : 1,
                                               h->act[L - 1], h->dact[L - 1], nullptr, loss_dst,
                                               at(h->grads, h->w_off[L - 1], h->esz), at(h->grads, h->b_off[L - 1], h->esz),
                                               at(h->grads, h->w_off[L - 2], h->esz), at(h->grads, h->b_off[L - 2], h->esz),
                                               h->dact[L - 3], h->dtype, h->pows, h->b1, h->b2));
        MLP_TRY(mlp_backward_layers(h, x, rows, L - 3, 1));
        const int64_t rest = h->w_off[1];
        STEP_CALL(h, tnn_dense_bwd_first_adam(rows, h->w[0], h->w[1], x, h->dact[0],
                                              h->keep_grads ? at(h->grads, h->w_off[0], h->esz) : nullptr,
                                              at(h->grads, h->b_off[0], h->esz), at(h->params, h->w_off[0], h->esz),
                                              at(h->m, h->w_off[0], h->esz), at(h->v, h->w_off[0], h->esz),
                                              at(h->params, h->b_off[0], h->esz), at(h->m, h->b_off[0], h->esz),
                                              at(h->v, h->b_off[0], h->esz), at(h->params, rest, h->esz),
                                              at(h->grads, rest, h->esz), at(h->m, rest, h->esz),
                                              at(h->v, rest, h->esz), h->n_params - rest, h->lr, h->b1, h->b2, h->eps,
                                              h->pows, h->dtype));
        return 0;
    }
    if (h->opt_kind == 1 && head_fits_one_workgroup(h, rows)) {
        // forward | loss (+ Adam's beta powers advanced by its thread 0) | backward, the last launch of which also
        // carries the optimizer
        MLP_TRY(mlp_forward(h, x, rows));
        STEP_CALL(h, tnn_softmax_nll_fused_tick(h->act[L - 1], y, rows, h->w[L], rows, 0, h->stats, loss_dst,
                                                h->dact[L - 1], h->dtype, h->pows, h->b1, h->b2));
        // backward of layers L-1 .. 1, then the first layer's backward with the whole Adam step folded into its
        // launch (its own W / b in the dW epilogue, every other layer's parameters by trailing blocks): 2L + 1 launches
        MLP_TRY(mlp_backward_layers(h, x, rows, -1, 1));
        const int64_t rest = L > 1 ? h->w_off[1] : h->n_params;
        STEP_CALL(h, tnn_dense_bwd_first_adam(rows, h->w[0], h->w[1], x, h->dact[0],
                                              h->keep_grads ? at(h->grads, h->w_off[0], h->esz) : nullptr,
                                              at(h->grads, h->b_off[0], h->esz), at(h->params, h->w_off[0], h->esz),
                                              at(h->m, h->w_off[0], h->esz), at(h->v, h->w_off[0], h->esz),
                                              at(h->params, h->b_off[0], h->esz), at(h->m, h->b_off[0], h->esz),
                                              at(h->v, h->b_off[0], h->esz), at(h->params, rest, h->esz),
                                              at(h->grads, rest, h->esz), at(h->m, rest, h->esz),
                                              at(h->v, rest, h->esz), h->n_params - rest, h->lr, h->b1, h->b2, h->eps,
                                              h->pows, h->dtype));
        return 0;
    }
    // hidden layers forward; then the classifier head (last Dense forward + loss + its backward: tnn_mlp_head);
    // then one launch per remaining layer backward; then the optimizer
    MLP_TRY(mlp_forward(h, x, rows, L - 1));
    STEP_CALL(h, tnn_mlp_head(rows, h->w[L - 1], h->w[L], L > 1 ? h->act[L - 2] : x,
                              at(h->params, h->w_off[L - 1], h->esz), at(h->params, h->b_off[L - 1], h->esz), y,
                              h->act[L - 1], h->dact[L - 1], h->stats, loss_dst,
                              at(h->grads, h->w_off[L - 1], h->esz), at(h->grads, h->b_off[L - 1], h->esz),
                              L > 1 ? h->dact[L - 2] : nullptr, h->dtype));
    MLP_TRY(mlp_backward_layers(h, x, rows, L - 2));
    return mlp_update(h);
}

static int step_sharded_impl(void* handle, const void* x, const void* y, int64_t rows, void* loss_out);

int tnn_mlp_step_sharded(void* handle, const void* x, const void* y, int64_t rows, void* loss_out) {
    // every update launched inside is tied to the peer-to-peer transport's health: if a peer barrier timed out, the
    // collectives were discarded on the device and so are the optimizer launches behind them (tnn_p2p_guard_updates)
    (void)tnn_p2p_guard_updates(1);
    const int rc = step_sharded_impl(handle, x, y, rows, loss_out);
    (void)tnn_p2p_guard_updates(0);
    return rc;
}

static int step_sharded_impl(void* handle, const void* x, const void* y, int64_t rows, void* loss_out) {
    // One data-parallel step, every phase enqueued from here on the library stream (no host work in between):
    //   forward + shard {max, sum-exp}  ->  C2 all-gather + log-sum-exp merge  ->  loss + backward with the GLOBAL
    //   batch size  ->  C1 in-place all-reduce of grads[0 : n_params + 1] (the loss rides along)  ->  update
    Mlp* h = (Mlp*)handle;
    MLP_TRY(check_rows(h, rows, "tnn_mlp_step_sharded"));
    h->call_idx = 0;
    int rank = 0, world = 1;
    MLP_TRY(tnn_comm_world(&rank, &world));
    if (h->stats_all_world != world) {
        tnn_free(h->stats_all);
        h->stats_all = nullptr;
        MLP_TRY(tnn_malloc((size_t)world * 2 * 8, &h->stats_all));
        h->stats_all_world = world;
    }
    if (zero_step_fits(h, world)) return mlp16_step_zero(h, x, y, rows, rank, world, loss_out);
    int p2p_on = 0;
    MLP_TRY(tnn_p2p_status(nullptr, &p2p_on, nullptr));
    const int L = h->L;
    void* loss_slot = at(h->grads, h->n_params, h->esz);
    // (the tuned 128 -> 10 head or any head of the generic merged kernel; <= 128 rows per rank in one block, 129 .. 1024 in
    // blocks of 128)
    // large arenas (config C: 134 MB, config E: 1 GB of fp32 gradients) take the per-layer buckets further down whatever
    // their head: one all-reduce per layer on the communication stream, overlapping the remaining backward
    // (TNN_BUCKET_BYTES moves the switch-over point; default 4 MiB — below it the arena is one latency-bound message)
    static const size_t bucket_bytes = getenv("TNN_BUCKET_BYTES") ? (size_t)atoll(getenv("TNN_BUCKET_BYTES")) : ((size_t)4 << 20);
    const bool bucketed = (size_t)(h->n_params + 1) * h->esz > bucket_bytes;
    if (!bucketed && head_fits_sharded(h, rows)) {
        // Classifier head of the one-launch form (<= 128 rows per rank: every weak-scaling point, config D at 8 ranks) —
        // 2L - 1 launches (5 for the MNIST net) + the collectives, ONE form for every transport:
        //   forward of the hidden layers; the LAST workgroup of the last one to finish also reduces the shard's {max, sum-exp}
        //   from the partial logits and, on the peer-to-peer transport, exchanges and merges them (tnn_dense_fwd_head_
        //   partials_stats: no statistics launch, nobody waits for a peer inside the head launch, no residency requirement)
        //   | RCCL only: tnn_allgather of the pairs | head + hidden layer's backward taking the pair(s) from memory
        //   (tnn_mlp_head_bwd_tick_ext; advances Adam's beta powers) | remaining backward | all-reduce + Adam
        MLP_TRY(mlp_forward(h, x, rows, L - 2));
        // more than 128 rows of the tuned 128 -> 10 head: the hidden layer's forward in its ROW-PANEL form (a workgroup finishes
        // 16 whole rows: logits + their statistics), the panels' pairs merged [+ exchanged] by the last workgroup of the launch
        // (tnn_dense_fwd_rows_head_stats_merged; pairs live behind the merged one in h->stats); otherwise the tiled forward whose
        // tail re-reads the partial logits per 128-row block behind arrival counters
        const bool row_panels = rows > 128 && h->w[L - 1] == 128 && h->w[L] == 10 && h->w[L - 2] % 4 == 0;
        // Peer-to-peer transport (round 6): the statistics exchange is DEFERRED into the head launch — the forward launch is the
        // single-GPU one (no acknowledgement wait, arrival ticket, system-scope re-read of the partial logits or exchange at its
        // tail: 8.4 -> 5.2 us at 128 rows), the head launch's workgroups reduce the shard's pair as they do on one GPU, one of
        // them pushes it to the peers and all merge the ranks' pairs from their own tagged slots (tnn_mlp_head_bwd_tick_xchg).
        // <= 128 rows (any head the merged launch takes), or the tuned head's row-panel form above 128 rows; a generic head with
        // more than 128 rows keeps the counter tail.  TNN_DP_XCHG=0 selects the round-5 form (A/B measurements).
        // (every workgroup of that head launch waits for the peers: tnn_mlp_head_bwd_xchg_fits also checks that the ranks whose
        // launches share THIS GPU — the tests' groups; never more than one in a one-process-per-GPU job — fit it together)
        static const bool xchg_allowed = !(getenv("TNN_DP_XCHG") && atoi(getenv("TNN_DP_XCHG")) == 0);
        int xchg_fits = 0;
        if (p2p_on && xchg_allowed && (rows <= 128 || row_panels))
            MLP_TRY(tnn_mlp_head_bwd_xchg_fits(rows, h->w[L - 2], h->w[L - 1], h->w[L], h->dtype, &xchg_fits));
        if (xchg_fits) {
            const int n_panels = (int)((rows + 15) / 16);
            if (row_panels)
                MLP_TRY(tnn_dense_fwd_rows_head_stats_merged(rows, h->w[L - 1], h->w[L - 2], h->act[L - 3], h->w[L - 2],
                                                             at(h->params, h->w_off[L - 2], h->esz), h->w[L - 1],
                                                             at(h->params, h->b_off[L - 2], h->esz), TNN_ACT_RELU, 1, h->act[L - 2],
                                                             h->w[L - 1], at(h->params, h->w_off[L - 1], h->esz), h->w[L], h->zpart,
                                                             at(h->params, h->b_off[L - 1], h->esz), (char*)h->stats + 16, h->ticket,
                                                             h->stats, 2, h->dtype));
            else
                MLP_TRY(tnn_dense_fwd_head_partials_stats(rows, h->w[L - 1], h->w[L - 2], h->act[L - 3], h->w[L - 2],
                                                          at(h->params, h->w_off[L - 2], h->esz), h->w[L - 1],
                                                          at(h->params, h->b_off[L - 2], h->esz), TNN_ACT_RELU, 1, h->act[L - 2],
                                                          h->w[L - 1], at(h->params, h->w_off[L - 1], h->esz), h->w[L], h->zpart,
                                                          at(h->params, h->b_off[L - 1], h->esz), y, h->ticket, h->stats, 2, h->dtype));
            MLP_TRY(tnn_mlp_head_bwd_tick_xchg(rows, rows * world, h->w[L - 2], h->w[L - 1], h->w[L], h->act[L - 3],
                                               at(h->params, h->w_off[L - 2], h->esz), h->act[L - 2],
                                               at(h->params, h->w_off[L - 1], h->esz), at(h->params, h->b_off[L - 1], h->esz),
                                               y, h->zpart, row_panels ? (char*)h->stats + 16 : nullptr, row_panels ? -n_panels : 0,
                                               h->act[L - 1], h->dact[L - 1], nullptr, loss_slot,
                                               at(h->grads, h->w_off[L - 1], h->esz), at(h->grads, h->b_off[L - 1], h->esz),
                                               at(h->grads, h->w_off[L - 2], h->esz), at(h->grads, h->b_off[L - 2], h->esz),
                                               h->dact[L - 3], h->dtype, h->pows, h->b1, h->b2));
            MLP_TRY(mlp_backward_layers(h, x, rows, L - 3, 1));
            return tnn_dense_bwd_first_allreduce_adam(rows, h->w[0], h->w[1], x, h->dact[0], h->grads, h->n_params + 1,
                                                      h->w_off[0], h->b_off[0], h->params, h->m, h->v, h->n_params, h->lr,
                                                      h->b1, h->b2, h->eps, h->pows, h->n_params, loss_out, h->dtype);
        }
        if (row_panels)
            MLP_TRY(tnn_dense_fwd_rows_head_stats_merged(rows, h->w[L - 1], h->w[L - 2], h->act[L - 3], h->w[L - 2],
                                                         at(h->params, h->w_off[L - 2], h->esz), h->w[L - 1],
                                                         at(h->params, h->b_off[L - 2], h->esz), TNN_ACT_RELU, 1, h->act[L - 2],
                                                         h->w[L - 1], at(h->params, h->w_off[L - 1], h->esz), h->w[L], h->zpart,
                                                         at(h->params, h->b_off[L - 1], h->esz), (char*)h->stats + 16, h->ticket,
                                                         h->stats, p2p_on ? 1 : 0, h->dtype));
        else
            MLP_TRY(tnn_dense_fwd_head_partials_stats(rows, h->w[L - 1], h->w[L - 2], h->act[L - 3], h->w[L - 2],
                                                      at(h->params, h->w_off[L - 2], h->esz), h->w[L - 1],
                                                      at(h->params, h->b_off[L - 2], h->esz), TNN_ACT_RELU, 1, h->act[L - 2],
                                                      h->w[L - 1], at(h->params, h->w_off[L - 1], h->esz), h->w[L], h->zpart,
                                                      at(h->params, h->b_off[L - 1], h->esz), y, h->ticket, h->stats,
                                                      p2p_on ? 1 : 0, h->dtype));
        const void* pairs = h->stats;
        int n_pairs = 1;
        if (!p2p_on) {
            MLP_TRY(tnn_allgather(h->stats, h->stats_all, 2, h->dtype));
            pairs = h->stats_all;
            n_pairs = world;
        }
        if (row_panels) n_pairs = -n_pairs;              // whole logits (without the bias) instead of per-tile partials
        MLP_TRY(tnn_mlp_head_bwd_tick_ext(rows, rows * world, h->w[L - 2], h->w[L - 1], h->w[L], h->act[L - 3],
                                          at(h->params, h->w_off[L - 2], h->esz), h->act[L - 2],
                                          at(h->params, h->w_off[L - 1], h->esz), at(h->params, h->b_off[L - 1], h->esz),
                                          y, h->zpart, pairs, n_pairs, h->act[L - 1], h->dact[L - 1], nullptr, loss_slot,
                                          at(h->grads, h->w_off[L - 1], h->esz), at(h->grads, h->b_off[L - 1], h->esz),
                                          at(h->grads, h->w_off[L - 2], h->esz), at(h->grads, h->b_off[L - 2], h->esz),
                                          h->dact[L - 3], h->dtype, h->pows, h->b1, h->b2));
        // remaining backward; the first layer's in one launch with the gradient all-reduce and Adam (peer-to-peer transport:
        // its tiles go straight into the owners' receive slots; RCCL: the launches that replaces)
        MLP_TRY(mlp_backward_layers(h, x, rows, L - 3, 1));
        return tnn_dense_bwd_first_allreduce_adam(rows, h->w[0], h->w[1], x, h->dact[0], h->grads, h->n_params + 1,
                                                  h->w_off[0], h->b_off[0], h->params, h->m, h->v, h->n_params, h->lr,
                                                  h->b1, h->b2, h->eps, h->pows, h->n_params, loss_out, h->dtype);
    }
    if (!bucketed && p2p_on && h->dtype == TNN_F32 && h->opt_kind == 1 && head_fits_one_workgroup(h, rows)) {
        // xGMI peer-to-peer transport and a head that fits one workgroup — 8 launches, like the single-GPU step:
        //   forward | loss kernel that exchanges the shards' {max, sum-exp} itself (stats + C2 + merge + loss + dz)
        //   and advances Adam's beta powers | backward | all-reduce whose last stage applies Adam and files the loss
        MLP_TRY(mlp_forward(h, x, rows));
        MLP_TRY(tnn_softmax_nll_fused_tick(h->act[L - 1], y, rows, h->w[L], rows * world, 1, h->stats, loss_slot,
                                           h->dact[L - 1], h->dtype, h->pows, h->b1, h->b2));
        MLP_TRY(mlp_backward_layers(h, x, rows));
        return tnn_allreduce_adam(h->grads, h->n_params + 1, h->params, h->m, h->v, h->n_params, h->lr, h->b1, h->b2,
                                  h->eps, h->pows, 0, h->dtype, h->n_params, loss_out);
    }
    MLP_TRY(mlp_forward_stats(h, x, rows, nullptr));
    if (h->loss_kind == 0) {
        MLP_TRY(tnn_allgather(h->stats, h->stats_all, 2, h->dtype));
        MLP_TRY(tnn_lse_merge(h->stats_all, world, h->stats, h->dtype));
    }
    if (bucketed) {
        // any failure from here to the optimizer drains the bucket events already issued (tnn_comm_join) before the
        // error is returned, so the next step never waits on this one's leftovers
        int rc = mlp_backward_impl(h, x, y, rows, rows * world, h->stats, nullptr, true);
        if (rc) { (void)tnn_comm_join(); return rc; }
        if (h->opt_kind == 1) {
            // Adam layer by layer in bucket order (last layer first): layer l's update starts when ITS bucket has
            // landed, while the earlier layers' buckets are still on the links
            for (int l = h->L - 1; l >= 0; --l) {
                rc = tnn_comm_wait_oldest();
                if (!rc) rc = adam_layer(h, l, l == h->L - 1, l == h->L - 1 ? loss_out : nullptr);
                if (rc) { (void)tnn_comm_join(); return rc; }
            }
            return 0;
        }
        MLP_TRY(tnn_comm_join());
        MLP_TRY(mlp_update(h));
        if (loss_out) MLP_TRY(tnn_memcpy_d2d(loss_out, loss_slot, h->esz));
        return 0;
    }
    MLP_TRY(mlp_backward_impl(h, x, y, rows, rows * world, h->stats, nullptr, false));
    if (!h->bf16 && h->opt_kind == 1)
        return tnn_allreduce_adam(h->grads, h->n_params + 1, h->params, h->m, h->v, h->n_params, h->lr, h->b1, h->b2,
                                  h->eps, h->pows, 1, h->dtype, h->n_params, loss_out);
    // the arenas of a bf16 trainer are fp32 (master weights, gradients, optimizer state)
    MLP_TRY(tnn_allreduce(h->grads, h->n_params + 1, h->bf16 ? TNN_F32 : h->dtype, TNN_RSUM));
    MLP_TRY(mlp_update(h));
    if (loss_out) MLP_TRY(tnn_memcpy_d2d(loss_out, loss_slot, h->esz));
    return 0;
}

int tnn_mlp_launch_window(void* handle, int first, int count, int* calls_in_last_step) {
    // Measurement hook: from now on tnn_mlp_step executes only its primitive calls number [first, first + count) (in
    // issue order, one launch each for the MNIST-size step); count < 0 restores the whole step.  While a window is set
    // only tnn_mlp_step may be called on this handle.  calls_in_last_step: how many primitive calls the last
    // tnn_mlp_step went through (executed or skipped) = launches per step.
    Mlp* h = (Mlp*)handle;
    if (!h || (count >= 0 && first < 0)) { tnn::set_error("tnn_mlp_launch_window: bad arguments"); return 2; }
    if (calls_in_last_step) *calls_in_last_step = (int)h->call_idx;
    h->windowed = count >= 0;
    h->win_lo = count < 0 ? 0 : first;
    h->win_hi = count < 0 ? 0 : first + count;
    return 0;
}

int tnn_mlp_keep_grads(void* handle, int keep) {
    Mlp* h = (Mlp*)handle;
    if (!h) { tnn::set_error("tnn_mlp_keep_grads: NULL handle"); return 2; }
    h->keep_grads = keep != 0;
    return 0;
}

int tnn_mlp_sync_params(void* handle) {
    Mlp* h = (Mlp*)handle;
    if (!h) { tnn::set_error("tnn_mlp_sync_params: NULL handle"); return 2; }
    h->masters_world = 0;          // the caller has just written the whole fp32 arena (initial weights, a checkpoint)
    return h->bf16 ? mlp16_sync(h) : 0;
}

int tnn_mlp_masters_sharded(void* handle, int* world) {
    Mlp* h = (Mlp*)handle;
    if (!h || !world) { tnn::set_error("tnn_mlp_masters_sharded: NULL argument"); return 2; }
    *world = h->masters_world;
    return 0;
}

int tnn_mlp_gather_masters(void* handle) {
    Mlp* h = (Mlp*)handle;
    if (!h) { tnn::set_error("tnn_mlp_gather_masters: NULL handle"); return 2; }
    return h->bf16 ? mlp16_gather_masters(h) : 0;
}

int tnn_mlp_bf16_weights(void* handle, void** w16) {
    Mlp* h = (Mlp*)handle;
    if (!h || !w16 || !h->bf16) { tnn::set_error("tnn_mlp_bf16_weights: not a bf16 trainer"); return 2; }
    if (h->w16_first_stale)          // (stays set: a recorded step replays without coming back through the host code)
        if (int rc = tnn_cast_bf16((float*)h->params + h->w_off[0], h->w16 + 2 * h->w_off[0], h->w[0] * h->w[1], 1)) return rc;
    *w16 = h->w16;
    return 0;
}

int tnn_mlp_activation(void* handle, int layer, void** ptr) {
    Mlp* h = (Mlp*)handle;
    if (!h || layer < 0 || layer >= h->L || !ptr) {
        tnn::set_error("tnn_mlp_activation: bad arguments");
        return 2;
    }
    *ptr = h->act[layer];
    return 0;
}

}  // extern "C"
