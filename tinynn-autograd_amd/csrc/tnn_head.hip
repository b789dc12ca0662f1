// Classifier head kernels of the MNIST-size step (K8 + K9): everything that touches the narrow last Dense layer in as few
// launches as possible, because each piece is far below a microsecond of math and would otherwise pay a kernel boundary each
// (SURVEY H3):
//     z  = a W + b                                  core/layers.py:49        (forward of the last Dense)
//     M, S, loss, dz = whole-batch softmax NLL      core/losses.py:24-32
//     dW = a^T dz,  db = column-sum dz              core/ops.py:159-160, :52-54
//     da = (dz W^T) * [pre-activation >= 0]         core/ops.py:156-157, :342-343 (mask = sign bit of a)
// History: a ONE-workgroup fusion of all of it (1024 threads, everything in LDS) measured 14.7 us against 2.4 + 5.3 + 2.8 us
// for three multi-block launches — a single workgroup pays every phase's LDS / barrier / load latency serially on one CU —
// and was removed in round 3; the MULTI-workgroup kernels below are what the step uses.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "tnn_internal.h"
#include "tnn_p2p.h"
#include <mutex>
#include <unordered_map>

#include "tnn_head_stats.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// MULTI-WORKGROUP head: the last Dense forward, the whole-batch softmax NLL and the last Dense backward of the
// single-GPU MNIST-size step in ONE launch WITHOUT serialising on one CU (the single-workgroup kernel above: 14.7 us).
// The head's inputs are so small — 128 x 128 activations, 128 x 10 weights — that EVERY workgroup recomputes the
// logits and the loss statistics for itself (164 k FMAs = 0.5 us of one CU's VALU time, all workgroups in parallel)
// and then produces only its own share of the outputs:
//     all G = H / 8 workgroups : z = a W + b, {M, S, L} -> dz [m, C] (kept in LDS)
//     workgroup g              : da rows [g * rpb, (g + 1) * rpb) = (dz W^T) * !signbit(a);  dW rows [8 g, 8 g + 8) = a^T dz
//     workgroup 0              : loss, stats, db = column sums of dz, Adam's beta powers advanced (tick)
//     workgroup G - 1          : logits (and dz, when asked for) written out
// so the step's forward-of-last-layer / loss / backward-of-last-layer triple (three launches, ~9.4 us of launch
// boundaries and first-load round trips around ~1 us of math) becomes one launch.  (JOURNAL.md, round 1, named this.)
//   * logits: thread (row r = t & 127, K-quarter kq = t >> 7) holds its 32 activations in registers (8 x 16-B loads,
//     requested first thing) and multiplies them with W rows read through the SCALAR cache (kq is wave-uniform, so
//     W[k][c] is an SGPR operand of v_fma: no LDS or vector traffic for W); the four K-quarter partials meet in LDS;
//   * statistics: one thread per row (waves 0-1), per-row normalisation against the row's own maximum, then ONE
//     combined reduction of {max, rescaled sum-exp, sum(log u + max)} by DPP wave reductions (tnn_internal.h) and a
//     2-entry LDS exchange — the arithmetic of nll_rows_body (tnn_nll_rows.h), so the numbers are the loss kernel's;
//   * da: thread (column j, row group) with W[j][:] in registers and dz rows broadcast from LDS; dW: 8 x C outputs per
//     workgroup, 4 row-quarters per output summed with two quad-permute DPP steps.
struct HeadMArgs {
    int m, rpb;                      // rows (<= 128; SH == 2: any number, walked in blocks of 128); da rows per workgroup
    int vec;                         // rows even, zpart / y 16-B aligned: the partial logits are staged with 16-B loads
    int nparts;                      // tiles of zpart that exist: H / 16 partial sums, or 1 = whole logits (bias still to be added)
    int m_global;                    // data-parallel (SH kernels): rows of the GLOBAL batch; the loss written is this rank's share
    const float* ext_pairs;          // SH >= 2: {M_q, S_q} pairs from an earlier launch (head_stats_kernel [+ all-gather])
    int ext_n;
    const tnn::p2p::XchgCtx* xc;     // SH == 3: the peer-to-peer group this launch exchanges the shard's pair with (device memory)
    int xw;                          // ... and its size (by value: a one-rank group must not cost a load through xc)
    const float *a, *w, *b, *y;
    const float* zpart;              // [H / 16][m][C] partial logits from the previous layer's tiles, or NULL
    float *logits, *dz, *stats, *loss, *dw, *db, *da;
    double* tick;
    double b1, b2;
};

// Partial logits -> logits, staged through LDS.  The [NP][m][C] partial array and the [m][C] labels are flat and contiguous,
// so thread f < m C / 4 sums its 16-B piece of every tile's array (NP + 1 fully coalesced loads) instead of each
// (row, class-triple) thread gathering 3 x NP scalars (24 vector-memory instructions per thread, ~6 cache lines each).
// RQ: the request (registers), ST: sum + bias -> zs / ys.  The caller puts its other loads between the two and a barrier
// after.  Odd row counts / unaligned labels take the element-wise loop.
template <int C, int NP>
struct HeadStage {
    f32x4 v[NP], yv;
};
template <int C, int NP>
__device__ __forceinline__ void head_stage_request(const HeadMArgs& p, const int t, HeadStage<C, NP>& h) {
    const int n = p.m * C;
    if (p.vec && t < (n >> 2)) {
#pragma unroll
        for (int tn = 0; tn < NP; ++tn) h.v[tn] = *reinterpret_cast<const f32x4*>(p.zpart + (size_t)tn * n + 4 * t);
        h.yv = *reinterpret_cast<const f32x4*>(p.y + 4 * t);
    }
}
template <int C, int NP>
__device__ __forceinline__ void head_stage_store(const HeadMArgs& p, const int t, const HeadStage<C, NP>& h, float* zs, float* ys) {
    static_assert(NP == 8, "the partial-sum tree is written for 8 tiles");
    const int n = p.m * C;
    if (p.vec) {
        if (t < (n >> 2)) {
            f32x4 s = ((h.v[0] + h.v[1]) + (h.v[2] + h.v[3])) + ((h.v[4] + h.v[5]) + (h.v[6] + h.v[7]));
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += p.b[(4 * t + i) % C];
            *reinterpret_cast<f32x4*>(zs + 4 * t) = s;
            *reinterpret_cast<f32x4*>(ys + 4 * t) = h.yv;
        }
    } else {
        for (int e = t; e < n; e += 512) {
            float u[NP];
#pragma unroll
            for (int tn = 0; tn < NP; ++tn) u[tn] = p.zpart[(size_t)tn * n + e];
            zs[e] = (((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]))) + p.b[e % C];
            ys[e] = p.y[e];
        }
    }
}

// PART: the logits arrive as H / 16 partial sums per element (tnn_dense_fwd_head_partials: the previous layer's 16-column
// tiles each contributed their share) and are only ADDED here; otherwise every workgroup computes them itself on
// v_mfma_f32_16x16x4_f32 (0.9 us of the CU's matrix pipe + the 64 KB activation read, measured).
// CUT (ablation builds of round 2, kept as a template parameter only): 0 = the kernel; 1 = stop after the logits, 2 = after
// the statistics, 3 = after dz.
// DA = false: the caller derives the hidden layer's dz itself (mlp_head_bwd_kernel below) — no da rows, no loads for them.
// SH (data parallel): 0 single GPU; 2 the shards' softmax statistics were reduced (and, on the peer-to-peer transport,
// exchanged and merged) at the tail of the previous launch (dense_fwd_head_kernel) and arrive through HeadMArgs::ext_pairs;
// 3 DEFERRED exchange (round 6, tnn_p2p.h: XchgCtx): every workgroup reduces the SHARD's pair itself exactly as for SH = 0,
// one workgroup of the launch pushes it to the peers and every workgroup merges the ranks' pairs from its own tagged slots —
// the forward launch in front has no statistics tail and nobody extends its exit by a link latency
template <int H, int C, bool PART, int CUT, bool DA, int SH = 0>
__device__ __forceinline__ void head_block(const HeadMArgs& p, const int g) {
    constexpr int ROWS = 128, ZS = C + 1, WS = 12, JPB = 8, G = H / JPB, KC = H / 16, NP = H / 16;
    static_assert(H == ROWS && C <= 12 && (H * C) % 4 == 0, "thread (t & 127) doubles as the hidden-unit index of the da phase");
    constexpr int TS = ROWS + 16;                  // row stride of the transposed images: (j or c, row chunk) -> distinct banks
    __shared__ float zs[ROWS * ZS];                // logits (MFMA form only); odd row stride: conflict-free with lane = row
    __shared__ __attribute__((aligned(16))) float ws[H * WS];      // W, rows padded to 12 (columns >= C hold 0)
    __shared__ __attribute__((aligned(16))) float dzr[ROWS * WS];  // dz [row][12]: the da phase reads whole rows (16-B broadcasts)
    __shared__ __attribute__((aligned(16))) float dzT[C * TS];     // dz^T [c][row]: the dW / db phases read 4 rows at a time
    __shared__ __attribute__((aligned(16))) float asT[JPB * TS];   // a[:, 8g : 8g + 8]^T [j][row] for this workgroup's dW rows
    __shared__ double red[8][4];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int i16 = lane & 15, grp = lane >> 4;
    const int r = t & (ROWS - 1), kq = t >> 7;     // da phase: column r, row group kq
    const int srow = t >> 2, sub = t & 3;          // statistics: row srow, classes sub, sub + 4, sub + 8
    const int m = p.m;
    const bool slive = srow < m;

    // ---- every global read of the kernel is requested here, before the first use, in as few vector-memory
    // instructions as possible; no exec-mask branches around loads: rows beyond m read a clamped (valid) address and
    // are neutralised in the arithmetic (their dz is 0, so whatever they loaded never reaches an output).
    const int sr = min(srow, m - 1);
    float zc[3] = {0.f, 0.f, 0.f}, yc[3];
    __shared__ __attribute__((aligned(16))) float zst[PART ? ROWS * C : 4], yst[PART ? ROWS * C : 4];   // staged logits / labels
    HeadStage<C, NP> stg;
    f32x4 av[PART ? 1 : KC];
    if constexpr (PART) {
        head_stage_request<C, NP>(p, t, stg);
    } else {
        const float* arow = p.a + (size_t)min(16 * wid + i16, m - 1) * H + 4 * grp;
#pragma unroll
        for (int c = 0; c < KC; ++c) av[c] = *reinterpret_cast<const f32x4*>(arow + 16 * c);
    }
    if constexpr (!PART) {
#pragma unroll
        for (int i = 0; i < 3; ++i) yc[i] = p.y[(size_t)sr * C + min(sub + 4 * i, C - 1)];
    }
    constexpr int WV = H * C / 4;                                       // float4 pieces of W (320)
    f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
    if (t < WV) w4 = *reinterpret_cast<const f32x4*>(p.w + 4 * t);
    float am[2] = {0.f, 0.f};
    const int r0 = g * p.rpb, rend = min(m, r0 + p.rpb);
    if constexpr (DA) {
#pragma unroll
        for (int i = 0; i < 2; ++i) am[i] = p.a[(size_t)min(r0 + kq + 4 * i, m - 1) * H + r];
    }
    const f32x4 asl = *reinterpret_cast<const f32x4*>(p.a + (size_t)min((t & 255) >> 1, m - 1) * H + g * JPB + 4 * (t & 1));
    // Adam's beta powers: read NOW (a dependent global round trip at the very end of workgroup 0 cost 1.3 us of the launch)
    double pw0 = 0.0, pw1 = 0.0;
    if (g == 0 && t == 0 && p.tick) { pw0 = p.tick[0]; pw1 = p.tick[1]; }

    // ---- W and the dW slice of a -> LDS (visible after the first barrier below)
    if (t < H) { ws[t * WS + 10] = 0.f; ws[t * WS + 11] = 0.f; }
    if (t < WV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int e = 4 * t + i; ws[(e / C) * WS + e % C] = w4[i]; }
    }
    if (t < 2 * ROWS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asT[(4 * (t & 1) + i) * TS + (t >> 1)] = (t >> 1) < m ? asl[i] : 0.f;
    }

    if constexpr (PART) {
        head_stage_store<C, NP>(p, t, stg, zst, yst);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            zc[i] = zst[sr * C + min(sub + 4 * i, C - 1)];
            yc[i] = yst[sr * C + min(sub + 4 * i, C - 1)];
        }
    } else {
        // wave w owns the 16-row tile [16 w, 16 w + 16) over the whole K = H: lane (i16, grp) holds a[row i16][16 c + 4 grp + j]
        // (registers, from global) and W[16 c + 4 grp + j][col i16] (LDS; columns 10, 11 are zeros, lanes i16 >= 12 re-read
        // column 11); two accumulator chains (40-cycle dependent latency against a 32-cycle issue)
        __syncthreads();
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* wl = ws + (4 * grp) * WS + min(i16, WS - 1);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][0], wl[(16 * c + 0) * WS], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][1], wl[(16 * c + 1) * WS], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][2], wl[(16 * c + 2) * WS], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][3], wl[(16 * c + 3) * WS], acc1, 0, 0, 0);
        }
        if (i16 < C) {
            const float bias = p.b[i16];
#pragma unroll
            for (int q = 0; q < 4; ++q) zs[(16 * wid + 4 * grp + q) * ZS + i16] = (acc0[q] + acc1[q]) + bias;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; ++i) zc[i] = zs[srow * ZS + min(sub + 4 * i, C - 1)];
    }
    if constexpr (CUT == 1) {
        p.da[(size_t)g * 512 + t] = zc[0] + zc[1] + zc[2] + am[0] + am[1] + yc[0];
        return;
    }

    HeadStats st;
    if constexpr (SH == 2) head_stats<C, true, true>(zc, yc, slive, sub, lane, wid, red, st, g == 0, p.ext_pairs, p.ext_n);
    else head_stats<C>(zc, yc, slive, sub, lane, wid, red, st, g == 0);     // only workgroup 0 writes the loss
    if constexpr (SH == 3) {                                                // the shard's pair -> the batch's (all ranks)
        __shared__ float xm[2];
        float Mx = st.M, Sx = (float)st.S;
        tnn::p2p::xchg_merge<512>(p.xc, p.xw, Mx, Sx, false, xm);
        st.M = Mx; st.S = (double)Sx;
    }
    const bool (&valid)[3] = st.valid;
    const float (&ec)[3] = st.ec, (&eyc)[3] = st.eyc;
    const float mx = st.mx, urow = st.urow;
    const float M = st.M;
    const double S = st.S;
    const double L = st.L;
    double inv_m = 1.0 / (double)m;
    if constexpr (SH != 0) inv_m = 1.0 / (double)p.m_global;
    if constexpr (CUT == 2) {
        p.da[(size_t)g * 512 + t] = (float)(S + L) + M + am[0] + am[1] + ec[0] + ec[1] + ec[2] + eyc[0];
        return;
    }
    float dzc[3];
    {
        // v_rcp_f32 (1 ulp) instead of two IEEE divisions on the way to dz (tolerance 1e-5)
        const float sf = slive ? expf(mx - M) * __builtin_amdgcn_rcpf((float)S) : 0.f, uf = slive ? (float)inv_m * __builtin_amdgcn_rcpf(urow) : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dzc[i] = ec[i] * sf - eyc[i] * uf;                  // 0 in the padding rows
            if (valid[i]) {
                dzr[srow * WS + sub + 4 * i] = dzc[i];
                dzT[(sub + 4 * i) * TS + srow] = dzc[i];
            }
        }
        if (sub >= 2) dzr[srow * WS + 8 + sub] = 0.f;          // columns 10, 11 of the 16-B row reads
    }
    __syncthreads();
    if constexpr (CUT == 3) {
        p.da[(size_t)g * 512 + t] = dzr[r * WS] + asT[r] + am[0] + am[1];
        return;
    }

    // ---- da rows of this workgroup: thread (column j = r, row group kq); W[j][:] and the dz row as three 16-B LDS reads
    // each (the dz row is wave-uniform: a broadcast)
    if (DA && p.da) {
        f32x4 wv[WS / 4];
#pragma unroll
        for (int i = 0; i < WS / 4; ++i) wv[i] = *reinterpret_cast<const f32x4*>(ws + r * WS + 4 * i);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + kq + 4 * i;
            if (row < rend) {
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < WS / 4; ++k) {
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(dzr + row * WS + 4 * k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d = fmaf(dv[e], wv[k][e], d);      // columns 10, 11: 0 * 0
                }
                p.da[(size_t)row * H + r] = (__float_as_uint(am[i]) >> 31) ? 0.f : d;
            }
        }
    }
    if constexpr (CUT == 4) return;
    // ---- dW rows [8g, 8g + 8): output o = (jl, c); lane q of a quad takes the 4-row chunks q, q + 4, q + 8 ... of the
    // transposed images (two 16-B reads per 4 FMAs; neighbouring lanes read neighbouring 16-B pieces: conflict-free).
    // The first version read a[row][j] and dz[row][c] element by element: 64 LDS reads per thread, 0.7 us.
    if (t < JPB * C * 4) {                                      // 320 threads = 5 whole waves
        const int q = t & 3, o = t >> 2, jl = o / C, c = o - jl * C;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < ROWS / 16; ++i) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(asT + jl * TS + 16 * i + 4 * q);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dzT + c * TS + 16 * i + 4 * q);
            s0 = fmaf(a4[0], d4[0], s0); s1 = fmaf(a4[1], d4[1], s1);
            s0 = fmaf(a4[2], d4[2], s0); s1 = fmaf(a4[3], d4[3], s1);
        }
        float s = s0 + s1;
        s += tnn::dpp_move<0xB1, 0xf>(0.f, s);
        s += tnn::dpp_move<0x4E, 0xf>(0.f, s);
        if (q == 0) p.dw[(size_t)(g * JPB + jl) * C + c] = s;
    }
    if constexpr (CUT == 5) return;
    if (g == 0) {
        if (wid == 7) {                                         // db[c] = sum_r dz[r][c]: lane (c, row quarter)
            const int c = min(lane & 15, C - 1), rq = lane >> 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) acc += *reinterpret_cast<const f32x4*>(dzT + c * TS + 32 * rq + 4 * i);
            float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < C) p.db[lane] = s;
        }
        if (t == 0) {
            // data parallel: this rank's SHARE of the global loss (the all-reduce of the gradient arena sums the shares)
            if (p.loss) p.loss[0] = SH != 0 ? (float)((((double)logf((float)S) + (double)M) * (double)m - L) * inv_m)
                                       : (float)((double)logf((float)S) + (double)M - L * inv_m);
            if (p.stats) { p.stats[0] = M; p.stats[1] = (float)S; }
            if (p.tick) { p.tick[0] = pw0 * p.b1; p.tick[1] = pw1 * p.b2; }
        }
    }
    if (g == G - 1 && slive) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (valid[i]) {
                if (p.logits) p.logits[(size_t)srow * C + sub + 4 * i] = zc[i];
                if (p.dz) p.dz[(size_t)srow * C + sub + 4 * i] = dzc[i];
            }
    }
}

template <int H, int C, bool PART, int CUT = 0>
__global__ __launch_bounds__(512) void mlp_head_multi_kernel(HeadMArgs p) {
    head_block<H, C, PART, CUT, true>(p, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------------------
// Head + the hidden layer's backward in ONE launch (4-launch step: fwd0 | fwd1 + partial logits | THIS | bwd0 + Adam).
// The hidden layer's dz (rows x H) is   dz1 = (dz W2^T) * !signbit(a1)   with dz = the loss gradient of the rows x C logits:
// 10 FMAs per element once dz is known, and dz costs every workgroup the same ~1 us the head's workgroups already spend
// (sum 8 partial logits, whole-batch statistics).  So the tiles of the hidden layer's backward derive the slice of dz1
// they contract over THEMSELVES instead of waiting for a launch boundary behind the head:
//     blocks [0, G)                 the head's workgroups (head_block, no da rows): dW2, db2, loss, stats, beta powers, logits
//     blocks [G, G + n_dw)          dW1 tile (16 inputs x 16 hidden units) = x^T dz1[:, 16 units]   (K = rows), db1 = column sums
//     blocks [G + n_dw, ...)        dx tile (16 rows x 16 inputs) = (dz1[16 rows, :] W1^T) * !signbit(x)   (K = H)
// Tile products as in small_tile_fast (v_mfma_f32_16x16x4_f32, one 16-deep K chunk per wave, 8 waves, partials summed
// through LDS); the dz1 operand comes from LDS (panel computed in place), the other operand through loads requested first
// thing.  dz1 is never written to HBM.
struct HeadBwdArgs {
    const float* x;                  // [m][n_in]   the hidden layer's input (sign-encoded ReLU output of the layer before)
    const float* w1;                 // [n_in][H]
    float *dw1, *db1, *dx;           // [n_in][H], [H], [m][n_in]
    int n_in, tiles_in;              // tiles_in = n_in / 16
    int xcd;                         // XCD-aware tile order (workgroup b runs on XCD b % 8)
    int dx_wide;                     // dx tiles of 16 rows x 32 inputs (tiles_in even).  <= 128 rows: for the MNIST net 16 + 128 + 64 = 208
                                     // workgroups instead of 272 — the 16 that were a CU's SECOND workgroup ended 1.1 us behind the rest
                                     // (profiles/r06_stepA_stamps.txt).  Row-blocked kernel: half the dx workgroups for its two slots per CU
                                     // (1024 rows: 656 instead of 1168 workgroups, profiles/r06_head_rb_stamps.txt)
};

// CUT (ablation builds of round 2, template parameter only): tile roles stop after 1 = the logits, 2 = the statistics, 3 = dz, 4 = the dz1 panel.
#ifdef TNN_STEP_TRACE
__device__ unsigned long long g_step_trace_head[1024 * 4];       // tnn_internal.h: TNN_STEP_STAMP (kernel id 2 of the step)
#endif

template <int H, int C, int CUT = 0, int SH = 0>
__global__ __launch_bounds__(512) void mlp_head_bwd_kernel(HeadMArgs p, HeadBwdArgs q) {
    constexpr int ROWS = 128, WS = 12, NP = H / 16, G = H / 8, TH = H / 16, PS = H + 4;
    static_assert(H == 128 && NP == 8, "one 16-deep K chunk per wave, 8 waves");
    TNN_STEP_STAMP(g_step_trace_head, 0, 0);
    if ((int)blockIdx.x < G) {
        head_block<H, C, true, 0, false, SH>(p, (int)blockIdx.x);
        TNN_STEP_STAMP_ACKED(g_step_trace_head, 0, 3);            // (head workgroups: entry and end only)
        return;
    }
    __shared__ __attribute__((aligned(16))) float zs[ROWS * C], ys[ROWS * C];     // staged logits / labels
    __shared__ __attribute__((aligned(16))) float dzr[ROWS * WS];   // dz [row][12] (columns 10, 11 hold 0)
    __shared__ __attribute__((aligned(16))) float pan[16 * PS];     // dx tiles: dz1 [16 rows][H], stride H + 4
    __shared__ float redm[8][8][64];                                // (registers 4 .. 7: the second column half of a wide dx tile)
    __shared__ float bsum[8][64];
    __shared__ double red[8][4];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int i16 = lane & 15, grp = lane >> 4;
    const int srow = t >> 2, sub = t & 3;
    const int m = p.m;
    const bool slive = srow < m;
    const int sr = min(srow, m - 1);
    const int n_dw = q.tiles_in * TH;
    const int blk = (int)blockIdx.x - G;
    const bool is_dw = blk < n_dw;
    const bool wide = q.dx_wide != 0;                                 // block-uniform
    const int n_in = q.n_in;
    // tile coordinates.  dW: tm over the inputs, tn over the hidden units; dx: tm over the rows, tn over the inputs
    int tm, tn;
    if (is_dw) {
        if (q.xcd && q.tiles_in % 2 == 0) {          // XCD-aware order: XCD x takes inputs' half x % 2, units' quarter x / 2
            const int xcd = blk & 7, idx = blk >> 3, pm = q.tiles_in / 2;
            tm = (xcd & 1) * pm + idx % pm;
            tn = (xcd >> 1) * (TH / 4) + idx / pm;
        } else { tm = blk % q.tiles_in; tn = blk / q.tiles_in; }
    } else {
        const int b2 = blk - n_dw, tr = (m + 15) / 16;
        const int tcols = wide ? q.tiles_in / 2 : q.tiles_in;        // dx tile columns (16 or 32 inputs each)
        if (q.xcd && tr % 2 == 0 && tcols % 4 == 0 && n_dw % 8 == 0) {
            const int xcd = b2 & 7, idx = b2 >> 3, pm = tr / 2;
            tm = (xcd & 1) * pm + idx % pm;
            tn = (xcd >> 1) * (tcols / 4) + idx / pm;
        } else { tm = b2 % tr; tn = b2 / tr; }
    }
    const int m0 = tm * 16, n0 = tn * ((!is_dw && wide) ? 32 : 16);

    // ---- every global read, up front.  The dz1 slice of a tile is itself a 16x16x4 MFMA product
    //   P[row][unit] = sum_c dz[row][c] W2[unit][c]    (K = 12: 10 classes + 2 zero columns, 3 MFMAs per wave)
    // dW tile: wave w takes rows 16 w .. 16 w + 15 x units n0 .. n0 + 15; dx tile: rows m0 .. m0 + 15 x units 16 w .. 16 w + 15.
    // Lane (i16, grp) supplies B[k = grp][n = i16] = W2[unit i16][class 4 s + grp] and receives P[row 4 grp + r][unit i16],
    // r = 0..3 — for the dW tile that IS the B fragment of the tile product (b[j] = dz1[16 w + 4 grp + j][n0 + i16]).
    HeadStage<C, NP> stg;
    head_stage_request<C, NP>(p, t, stg);
    const int urow = (is_dw ? n0 : 16 * wid) + i16;                  // this lane's hidden unit
    const int prow0 = (is_dw ? 16 * wid : m0) + 4 * grp;             // first of this lane's 4 panel rows
    float w2f[3], a1m[4];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) w2f[s3] = p.w[(size_t)urow * C + min(4 * s3 + grp, C - 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) a1m[r] = p.a[(size_t)min(prow0 + r, m - 1) * H + urow];
    float af[4] = {0.f, 0.f, 0.f, 0.f};          // dW: x fragment
    f32x4 bf = {0.f, 0.f, 0.f, 0.f}, bf1 = {0.f, 0.f, 0.f, 0.f};     // dx: W1 fragment(s)
    float e_pre = 0.f, e_pre1 = 0.f;             // dx: mask source(s)
    const int e_r = (t >> 6) & 3, e_ln = t & 63;
    if (is_dw) {
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = q.x[(size_t)min(16 * wid + 4 * grp + j, m - 1) * n_in + m0 + i16];
    } else {
        bf = *reinterpret_cast<const f32x4*>(q.w1 + (size_t)(n0 + i16) * H + 16 * wid + 4 * grp);
        if (wide) bf1 = *reinterpret_cast<const f32x4*>(q.w1 + (size_t)(n0 + 16 + i16) * H + 16 * wid + 4 * grp);
        if (t < 256) {
            const float* xr = q.x + (size_t)min(m0 + (e_ln >> 4) * 4 + e_r, m - 1) * n_in + n0 + (e_ln & 15);
            e_pre = xr[0];
            if (wide) e_pre1 = xr[16];
        }
    }
    if (grp >= 2) w2f[2] = 0.f;                  // classes 10, 11 do not exist (the clamped address read class 9)
    static_assert(C == 10, "the zero columns of the K = 12 product are written for 10 classes");

    TNN_STEP_STAMP_ACKED(g_step_trace_head, 0, 1);
    head_stage_store<C, NP>(p, t, stg, zs, ys);
    __syncthreads();
    float zc[3], yc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        zc[i] = zs[sr * C + min(sub + 4 * i, C - 1)];
        yc[i] = ys[sr * C + min(sub + 4 * i, C - 1)];
    }
    if constexpr (CUT == 1) {
        q.dx[(size_t)(blk % 64) * 512 + t] = zc[0] + zc[1] + zc[2] + yc[0] + w2f[0] + w2f[1] + w2f[2] + a1m[0] + a1m[1] + a1m[2] + a1m[3] + af[0] + af[1] + af[2] + af[3] + bf[0] + e_pre;
        return;
    }
    HeadStats st;
    if constexpr (SH == 2) head_stats<C, false, true>(zc, yc, slive, sub, lane, wid, red, st, false, p.ext_pairs, p.ext_n);
    else head_stats<C, false>(zc, yc, slive, sub, lane, wid, red, st);
    if constexpr (SH == 3) {                       // the first tile workgroup is the launch's sender (f32 statistics: the shortest path to the pair)
        __shared__ float xm[2];
        float Mx = st.M, Sx = (float)st.S;
        tnn::p2p::xchg_merge<512>(p.xc, p.xw, Mx, Sx, blk == 0, xm);
        st.M = Mx; st.S = (double)Sx;
    }
    if constexpr (CUT == 2) {
        q.dx[(size_t)(blk % 64) * 512 + t] = (float)st.S + st.M + st.ec[0] + st.eyc[1] + w2f[0] + w2f[1] + w2f[2] + a1m[0] + a1m[1] + a1m[2] + a1m[3] + af[0] + af[1] + af[2] + af[3] + bf[0] + e_pre;
        return;
    }
    float m_norm = (float)m;
    if constexpr (SH != 0) m_norm = (float)p.m_global;
    {
        const float sf = slive ? expf(st.mx - st.M) * __builtin_amdgcn_rcpf((float)st.S) : 0.f;
        const float uf = slive ? __builtin_amdgcn_rcpf(m_norm * st.urow) : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (st.valid[i]) dzr[srow * WS + sub + 4 * i] = st.ec[i] * sf - st.eyc[i] * uf;      // 0 in the padding rows
        if (sub >= 2) dzr[srow * WS + 8 + sub] = 0.f;
    }
    // dW tile: wave w reads back only the 16 rows it wrote itself (srow = 16 w + lane / 4) — no workgroup barrier, the
    // LDS queue of a wave is in order; dx tile: the 16 rows of the tile were written by wave tm
    if (is_dw) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
    if constexpr (CUT == 3) {
        q.dx[(size_t)(blk % 64) * 512 + t] = dzr[t] + w2f[0] + w2f[1] + w2f[2] + a1m[0] + a1m[1] + a1m[2] + a1m[3] + af[0] + af[1] + af[2] + af[3] + bf[0] + e_pre;
        return;
    }

    f32x4 pz = {0.f, 0.f, 0.f, 0.f};
    {
        const float* drow = dzr + ((is_dw ? 16 * wid : m0) + i16) * WS + grp;          // A[m = i16][k = grp] = dz[row][4 s + grp]
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) pz = __builtin_amdgcn_mfma_f32_16x16x4f32(drow[4 * s3], w2f[s3], pz, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) pz[r] = (__float_as_uint(a1m[r]) >> 31) ? 0.f : pz[r];   // rows >= m: dz = 0 there
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bs = 0.f;
    if (is_dw) {
        if constexpr (CUT == 4) {
            q.dx[(size_t)(blk % 64) * 512 + t] = pz[0] + pz[1] + pz[2] + pz[3] + af[0] + af[1] + af[2] + af[3];
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], pz[j], acc, 0, 0, 0);
        bs = (pz[0] + pz[1]) + (pz[2] + pz[3]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) pan[(4 * grp + r) * PS + 16 * wid + i16] = pz[r];
        __syncthreads();
        if constexpr (CUT == 4) {
            q.dx[(size_t)(blk % 64) * 512 + t] = pan[t] + bf[0] + e_pre;
            return;
        }
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(pan + i16 * PS + 16 * wid + 4 * grp);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bf[j], acc, 0, 0, 0);
        if (wide) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bf1[j], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) redm[wid][r][lane] = acc[r];
    if (!is_dw && wide) {
#pragma unroll
        for (int r = 0; r < 4; ++r) redm[wid][4 + r][lane] = acc1[r];
    }
    bsum[wid][lane] = bs;
    __syncthreads();
    TNN_STEP_STAMP(g_step_trace_head, 0, 2);
    if (t < 256) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += redm[w][e_r][e_ln];
        const int row = m0 + (e_ln >> 4) * 4 + e_r, col = n0 + (e_ln & 15);           // 16x16x4 C/D layout
        if (is_dw) q.dw1[(size_t)row * H + col] = s;
        else if (row < m) {
            q.dx[(size_t)row * n_in + col] = (__float_as_uint(e_pre) >> 31) ? 0.f : s;
            if (wide) {
                float s1 = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) s1 += redm[w][4 + e_r][e_ln];
                q.dx[(size_t)row * n_in + col + 16] = (__float_as_uint(e_pre1) >> 31) ? 0.f : s1;
            }
        }
    }
    if (is_dw && tm == 0 && t < 16) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += (bsum[w][t] + bsum[w][16 + t]) + (bsum[w][32 + t] + bsum[w][48 + t]);
        q.db1[n0 + t] = s;
    }
    TNN_STEP_STAMP_ACKED(g_step_trace_head, 0, 3);
}

// ------------------------------------------------------------------------------------------------------------------
// mb / row0: the row block staged (rows [row0, row0 + mb) of the p.m rows the partial array was written for).
template <int C, int NP>
__device__ __forceinline__ void head_stage_request_rb(const HeadMArgs& p, const int t, HeadStage<C, NP>& h, const int mb, const int row0) {
    const int n = mb * C, stride = p.m * C, base = row0 * C;
    if (p.vec && t < (n >> 2)) {
#pragma unroll
        for (int tn = 0; tn < NP; ++tn) {
            h.v[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (tn < p.nparts) h.v[tn] = *reinterpret_cast<const f32x4*>(p.zpart + (size_t)tn * stride + base + 4 * t);   // uniform
        }
        h.yv = *reinterpret_cast<const f32x4*>(p.y + base + 4 * t);
    }
}
template <int C, int NP>
__device__ __forceinline__ f32x4 head_stage_bias(const HeadMArgs& p, const int t) {
    // the bias of this thread's four staged elements — the same for every row block (a block starts at a multiple of C floats)
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (p.vec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = p.b[(4 * t + i) % C];
    }
    return b;
}
template <int C, int NP>
__device__ __forceinline__ void head_stage_store_rb(const HeadMArgs& p, const int t, const HeadStage<C, NP>& h, float* zs, float* ys,
                                                 const int mb, const int row0, const f32x4 bias4) {
    static_assert(NP == 8, "the partial-sum tree is written for 8 tiles");
    const int n = mb * C, stride = p.m * C, base = row0 * C;
    if (p.vec) {
        if (t < (n >> 2)) {
            f32x4 s = ((h.v[0] + h.v[1]) + (h.v[2] + h.v[3])) + ((h.v[4] + h.v[5]) + (h.v[6] + h.v[7]));
            s += bias4;
            *reinterpret_cast<f32x4*>(zs + 4 * t) = s;
            *reinterpret_cast<f32x4*>(ys + 4 * t) = h.yv;
        }
    } else {
#pragma unroll 1                                   // the rare path (odd row count): keep its registers off the kernel's budget
        for (int e = t; e < n; e += 512) {
            float u[NP];
#pragma unroll
            for (int tn = 0; tn < NP; ++tn) u[tn] = tn < p.nparts ? p.zpart[(size_t)tn * stride + base + e] : 0.f;
            zs[e] = (((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]))) + p.b[e % C];
            ys[e] = p.y[base + e];
        }
    }
}

// ROW-BLOCKED form of head_block (statistics from memory, any number of rows walked in blocks of 128; launched for more than
// 128 rows only — the <= 128-row kernels above are the code tuned in round 2, untouched: sharing one body with the loop cost
// them 0.3 us of the 128-row step, measured against the library of the round's start).
// PART: the logits arrive as H / 16 partial sums per element (tnn_dense_fwd_head_partials: the previous layer's 16-column
// tiles each contributed their share) and are only ADDED here; otherwise every workgroup computes them itself on
// v_mfma_f32_16x16x4_f32 (0.9 us of the CU's matrix pipe + the 64 KB activation read, measured).
// CUT (ablation builds of round 2, kept as a template parameter only): 0 = the kernel; 1 = stop after the logits, 2 = after
// the statistics, 3 = after dz.
// DA = false: the caller derives the hidden layer's dz itself (mlp_head_bwd_kernel below) — no da rows, no loads for them.
// SH (data parallel): 0 single GPU; 2 the shards' softmax statistics were reduced (and, on the peer-to-peer transport,
// exchanged and merged) at the tail of the previous launch (dense_fwd_head_kernel) and arrive through HeadMArgs::ext_pairs
// RB (with SH == 2 only): the rows are walked in blocks of 128 (any number of rows); without it one block, as before
template <int H, int C, bool PART, int CUT, bool DA, int SH = 0, bool RB = false>
__device__ __forceinline__ void head_block_rb(const HeadMArgs& p, const int g) {
    constexpr int ROWS = 128, ZS = C + 1, WS = 12, JPB = 8, G = H / JPB, KC = H / 16, NP = H / 16;
    static_assert(H == ROWS && C <= 12 && (H * C) % 4 == 0, "thread (t & 127) doubles as the hidden-unit index of the da phase");
    static_assert(!RB || (SH >= 2 && PART && !DA && CUT == 0), "row blocks exist in the form that takes the statistics from memory only");
    constexpr int TS = ROWS + 16;                  // row stride of the transposed images: (j or c, row chunk) -> distinct banks
    __shared__ float zs[ROWS * ZS];                // logits (MFMA form only); odd row stride: conflict-free with lane = row
    __shared__ __attribute__((aligned(16))) float ws[H * WS];      // W, rows padded to 12 (columns >= C hold 0)
    __shared__ __attribute__((aligned(16))) float dzr[ROWS * WS];  // dz [row][12]: the da phase reads whole rows (16-B broadcasts)
    __shared__ __attribute__((aligned(16))) float dzT[C * TS];     // dz^T [c][row]: the dW / db phases read 4 rows at a time
    __shared__ __attribute__((aligned(16))) float asT[JPB * TS];   // a[:, 8g : 8g + 8]^T [j][row] for this workgroup's dW rows
    __shared__ double red[8][4];
    __shared__ __attribute__((aligned(16))) float zst[PART ? ROWS * C : 4], yst[PART ? ROWS * C : 4];   // staged logits / labels
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int i16 = lane & 15, grp = lane >> 4;
    const int r = t & (ROWS - 1), kq = t >> 7;     // da phase: column r, row group kq
    const int srow = t >> 2, sub = t & 3;          // statistics: row srow, classes sub, sub + 4, sub + 8
    // SH == 2 (statistics from memory: nothing couples the rows inside this launch any more): ANY number of rows, taken in
    // blocks of 128 whose contributions to dW / db / the loss are accumulated in registers; otherwise one block
    const int nb = RB ? (p.m + ROWS - 1) / ROWS : 1;

    // ---- the loads that do not depend on the row block
    constexpr int WV = H * C / 4;                                       // float4 pieces of W (320)
    f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
    if (t < WV) w4 = *reinterpret_cast<const f32x4*>(p.w + 4 * t);
    // Adam's beta powers: read NOW (a dependent global round trip at the very end of workgroup 0 cost 1.3 us of the launch)
    double pw0 = 0.0, pw1 = 0.0;
    if (g == 0 && t == 0 && p.tick) { pw0 = p.tick[0]; pw1 = p.tick[1]; }

    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PART) bias4 = head_stage_bias<C, NP>(p, t);
    float ext_ms[2] = {0.f, 0.f};                  // SH >= 2: the batch's {max, sum-exp}, merged once for all row blocks
    if constexpr (SH >= 2) head_merge_pairs(p.ext_pairs, p.ext_n, lane, ext_ms[0], ext_ms[1]);
    if constexpr (SH == 3) {                       // ... the pairs in memory were this SHARD's: exchange and merge with the peers'
        __shared__ float xm[2];
        tnn::p2p::xchg_merge<512>(p.xc, p.xw, ext_ms[0], ext_ms[1], false, xm);
    }
    float dws0 = 0.f, dws1 = 0.f;                  // this thread's dW partial sums (threads < JPB * C * 4)
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};            // workgroup 0, wave 7: db partial sums
    double Lsum = 0.0;
    float M = 0.f;
    double S = 0.0;
    // row blocks: the global reads of block rb + 1 are requested as soon as block rb's staged values sit in LDS (a block's
    // chain of loads -> LDS -> statistics -> products is ~1 us of dependent latency plus the loads' own 1.5-2 us, which this hides)
    HeadStage<C, NP> stg_nx;
    f32x4 asl_nx = {0.f, 0.f, 0.f, 0.f};
    constexpr bool PF = true;
    if constexpr (RB && PF) {
        const int m0b = min(ROWS, p.m);
        head_stage_request_rb<C, NP>(p, t, stg_nx, m0b, 0);
        asl_nx = *reinterpret_cast<const f32x4*>(p.a + (size_t)min((t & 255) >> 1, m0b - 1) * H + g * JPB + 4 * (t & 1));
    }
    for (int rb = 0; rb < nb; ++rb) {
    int row0 = rb * ROWS;
    // opaque to the optimiser: otherwise every global address of the block is strength-reduced into a loop-carried 64-bit
    // register pair (~50 VGPRs, the kernel drops to one workgroup per CU)
    if constexpr (RB) asm volatile("" : "+s"(row0));
    const int m = RB ? min(ROWS, p.m - row0) : p.m;
    const float* const a_rb = p.a + (size_t)row0 * H;
    const bool slive = srow < m;
    if (rb) __syncthreads();                       // every LDS image of the previous block has been consumed
    int tt = t;                                    // the thread index as the block's global addresses see it (opaque, see row0)
    if constexpr (RB) asm volatile("" : "+v"(tt));

    // ---- every global read of the block is requested here, before the first use, in as few vector-memory
    // instructions as possible; no exec-mask branches around loads: rows beyond m read a clamped (valid) address and
    // are neutralised in the arithmetic (their dz is 0, so whatever they loaded never reaches an output).
    const int sr = min(srow, m - 1);
    float zc[3] = {0.f, 0.f, 0.f}, yc[3];
    HeadStage<C, NP> stg;
    f32x4 av[PART ? 1 : KC];
    if constexpr (PART) {
        if constexpr (RB && PF) stg = stg_nx;
        else head_stage_request_rb<C, NP>(p, tt, stg, m, row0);
    } else {
        const float* arow = a_rb + (size_t)min(16 * wid + i16, m - 1) * H + 4 * grp;
#pragma unroll
        for (int c = 0; c < KC; ++c) av[c] = *reinterpret_cast<const f32x4*>(arow + 16 * c);
    }
    if constexpr (!PART) {
#pragma unroll
        for (int i = 0; i < 3; ++i) yc[i] = p.y[(size_t)(row0 + sr) * C + min(sub + 4 * i, C - 1)];
    }
    float am[2] = {0.f, 0.f};
    const int r0 = g * p.rpb, rend = min(m, r0 + p.rpb);
    if constexpr (DA) {
#pragma unroll
        for (int i = 0; i < 2; ++i) am[i] = a_rb[(size_t)min(r0 + kq + 4 * i, m - 1) * H + r];
    }
    f32x4 asl;
    if constexpr (RB && PF) asl = asl_nx;
    else asl = *reinterpret_cast<const f32x4*>(a_rb + (size_t)min((t & 255) >> 1, m - 1) * H + g * JPB + 4 * (t & 1));

    // ---- W and the dW slice of a -> LDS (visible after the first barrier below)
    if (rb == 0) {
        if (t < H) { ws[t * WS + 10] = 0.f; ws[t * WS + 11] = 0.f; }
        if (t < WV) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int e = 4 * t + i; ws[(e / C) * WS + e % C] = w4[i]; }
        }
    }
    if (t < 2 * ROWS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asT[(4 * (t & 1) + i) * TS + (t >> 1)] = (t >> 1) < m ? asl[i] : 0.f;
    }

    if constexpr (PART) {
        head_stage_store_rb<C, NP>(p, t, stg, zst, yst, m, row0, bias4);
        if constexpr (RB && PF) {
            // the staged values are in LDS, their registers are free: the next block's reads travel from here on
            if (rb + 1 < nb) {
                const int m1 = min(ROWS, p.m - row0 - ROWS);
                head_stage_request_rb<C, NP>(p, tt, stg_nx, m1, row0 + ROWS);
                asl_nx = *reinterpret_cast<const f32x4*>(a_rb + (size_t)(ROWS + min((tt & 255) >> 1, m1 - 1)) * H + g * JPB + 4 * (tt & 1));
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            zc[i] = zst[sr * C + min(sub + 4 * i, C - 1)];
            yc[i] = yst[sr * C + min(sub + 4 * i, C - 1)];
        }
    } else {
        // wave w owns the 16-row tile [16 w, 16 w + 16) over the whole K = H: lane (i16, grp) holds a[row i16][16 c + 4 grp + j]
        // (registers, from global) and W[16 c + 4 grp + j][col i16] (LDS; columns 10, 11 are zeros, lanes i16 >= 12 re-read
        // column 11); two accumulator chains (40-cycle dependent latency against a 32-cycle issue)
        __syncthreads();
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* wl = ws + (4 * grp) * WS + min(i16, WS - 1);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][0], wl[(16 * c + 0) * WS], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][1], wl[(16 * c + 1) * WS], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][2], wl[(16 * c + 2) * WS], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][3], wl[(16 * c + 3) * WS], acc1, 0, 0, 0);
        }
        if (i16 < C) {
            const float bias = p.b[i16];
#pragma unroll
            for (int q = 0; q < 4; ++q) zs[(16 * wid + 4 * grp + q) * ZS + i16] = (acc0[q] + acc1[q]) + bias;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; ++i) zc[i] = zs[srow * ZS + min(sub + 4 * i, C - 1)];
    }
    if constexpr (CUT == 1) {
        p.da[(size_t)g * 512 + t] = zc[0] + zc[1] + zc[2] + am[0] + am[1] + yc[0];
        return;
    }

    HeadStats st;
    // RB: the loss-writing workgroup keeps each row's log term in a register and reduces ONCE behind the walk — the per-block
    // reduction (a barrier + f64 DPP sums) made workgroup 0 the launch's last by 4 us at 1024 rows (17.2 against 13.1 for its peers)
    if constexpr (SH >= 2) head_stats<C, true, true>(zc, yc, slive, sub, lane, wid, red, st, !RB && g == 0, ext_ms, 0);
    else head_stats<C>(zc, yc, slive, sub, lane, wid, red, st, g == 0);     // only workgroup 0 writes the loss
    const bool (&valid)[3] = st.valid;
    const float (&ec)[3] = st.ec, (&eyc)[3] = st.eyc;
    const float mx = st.mx, urow = st.urow;
    M = st.M;
    S = st.S;
    if constexpr (RB) { if (g == 0 && slive && sub == 0) Lsum += (double)logf(urow) + (double)mx; }
    else Lsum += st.L;
    double inv_m = 1.0 / (double)m;
    if constexpr (SH != 0) inv_m = 1.0 / (double)p.m_global;
    if constexpr (CUT == 2) {
        p.da[(size_t)g * 512 + t] = (float)(S + st.L) + M + am[0] + am[1] + ec[0] + ec[1] + ec[2] + eyc[0];
        return;
    }
    float dzc[3];
    {
        // v_rcp_f32 (1 ulp) instead of two IEEE divisions on the way to dz (tolerance 1e-5)
        const float sf = slive ? expf(mx - M) * __builtin_amdgcn_rcpf((float)S) : 0.f, uf = slive ? (float)inv_m * __builtin_amdgcn_rcpf(urow) : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dzc[i] = ec[i] * sf - eyc[i] * uf;                  // 0 in the padding rows
            if (valid[i]) {
                dzr[srow * WS + sub + 4 * i] = dzc[i];
                dzT[(sub + 4 * i) * TS + srow] = dzc[i];
            }
        }
        if (sub >= 2) dzr[srow * WS + 8 + sub] = 0.f;          // columns 10, 11 of the 16-B row reads
    }
    __syncthreads();
    if constexpr (CUT == 3) {
        p.da[(size_t)g * 512 + t] = dzr[r * WS] + asT[r] + am[0] + am[1];
        return;
    }

    // ---- da rows of this workgroup: thread (column j = r, row group kq); W[j][:] and the dz row as three 16-B LDS reads
    // each (the dz row is wave-uniform: a broadcast)
    if (DA && p.da) {
        f32x4 wv[WS / 4];
#pragma unroll
        for (int i = 0; i < WS / 4; ++i) wv[i] = *reinterpret_cast<const f32x4*>(ws + r * WS + 4 * i);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + kq + 4 * i;
            if (row < rend) {
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < WS / 4; ++k) {
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(dzr + row * WS + 4 * k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d = fmaf(dv[e], wv[k][e], d);      // columns 10, 11: 0 * 0
                }
                p.da[(size_t)row * H + r] = (__float_as_uint(am[i]) >> 31) ? 0.f : d;
            }
        }
    }
    if constexpr (CUT == 4) return;
    // ---- dW rows [8g, 8g + 8): output o = (jl, c); lane q of a quad takes the 4-row chunks q, q + 4, q + 8 ... of the
    // transposed images (two 16-B reads per 4 FMAs; neighbouring lanes read neighbouring 16-B pieces: conflict-free).
    // The first version read a[row][j] and dz[row][c] element by element: 64 LDS reads per thread, 0.7 us.
    if (t < JPB * C * 4) {                                      // 320 threads = 5 whole waves
        const int q = t & 3, o = t >> 2, jl = o / C, c = o - jl * C;
#pragma unroll
        for (int i = 0; i < ROWS / 16; ++i) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(asT + jl * TS + 16 * i + 4 * q);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dzT + c * TS + 16 * i + 4 * q);
            dws0 = fmaf(a4[0], d4[0], dws0); dws1 = fmaf(a4[1], d4[1], dws1);
            dws0 = fmaf(a4[2], d4[2], dws0); dws1 = fmaf(a4[3], d4[3], dws1);
        }
    }
    if (g == 0 && wid == 7) {                                   // db[c] = sum_r dz[r][c]: lane (c, row quarter)
        const int c = min(lane & 15, C - 1), rq = lane >> 4;
#pragma unroll
        for (int i = 0; i < ROWS / 16; ++i) dbacc += *reinterpret_cast<const f32x4*>(dzT + c * TS + 32 * rq + 4 * i);
    }
    if (g == G - 1 && slive) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (valid[i]) {
                if (p.logits) p.logits[(size_t)(row0 + srow) * C + sub + 4 * i] = zc[i];
                if (p.dz) p.dz[(size_t)(row0 + srow) * C + sub + 4 * i] = dzc[i];
            }
    }
    }   // row blocks

    if (t < JPB * C * 4) {
        const int q = t & 3, o = t >> 2, jl = o / C, c = o - jl * C;
        float s = dws0 + dws1;
        s += tnn::dpp_move<0xB1, 0xf>(0.f, s);
        s += tnn::dpp_move<0x4E, 0xf>(0.f, s);
        if (q == 0) p.dw[(size_t)(g * JPB + jl) * C + c] = s;
    }
    if constexpr (CUT == 5) return;
    if constexpr (RB) {
        if (g == 0) {                                            // block-uniform: the rows' log terms, lanes -> waves -> thread 0
            const double wl = tnn::wave_sum_dpp(Lsum);
            __syncthreads();                                    // (red was last read inside the walk)
            if (lane == 0) red[wid][2] = wl;
            __syncthreads();
            Lsum = ((red[0][2] + red[1][2]) + (red[2][2] + red[3][2])) + ((red[4][2] + red[5][2]) + (red[6][2] + red[7][2]));
        }
    }
    if (g == 0) {
        if (wid == 7) {
            float s = (dbacc[0] + dbacc[1]) + (dbacc[2] + dbacc[3]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < C) p.db[lane] = s;
        }
        if (t == 0) {
            // data parallel: this rank's SHARE of the global loss (the all-reduce of the gradient arena sums the shares)
            const double inv_all = 1.0 / (double)(SH != 0 ? p.m_global : p.m);
            if (p.loss) p.loss[0] = SH != 0 ? (float)((((double)logf((float)S) + (double)M) * (double)p.m - Lsum) * inv_all)
                                       : (float)((double)logf((float)S) + (double)M - Lsum * inv_all);
            if (p.stats) { p.stats[0] = M; p.stats[1] = (float)S; }
            if (p.tick) { p.tick[0] = pw0 * p.b1; p.tick[1] = pw1 * p.b2; }
        }
    }
}

// CUT (ablation builds of round 2, template parameter only): tile roles stop after 1 = the logits, 2 = the statistics, 3 = dz, 4 = the dz1 panel.
template <int H, int C, int CUT = 0, int SH = 0, bool RB = false>
__global__ __launch_bounds__(512, RB ? 4 : 2) void mlp_head_bwd_rb_kernel(HeadMArgs p, HeadBwdArgs q) {
    constexpr int ROWS = 128, WS = 12, NP = H / 16, G = H / 8, TH = H / 16, PS = H + 4;
    static_assert(H == 128 && NP == 8, "one 16-deep K chunk per wave, 8 waves");
    TNN_STEP_STAMP(g_step_trace_head, 0, 0);                     // (trace build: entry / end of every workgroup, tools/probes/step_stamps.py)
    if ((int)blockIdx.x < G) {
        head_block_rb<H, C, true, 0, false, SH, RB>(p, (int)blockIdx.x);
        TNN_STEP_STAMP_ACKED(g_step_trace_head, 0, 3);
        return;
    }
    __shared__ __attribute__((aligned(16))) float zs[ROWS * C], ys[ROWS * C];     // staged logits / labels
    __shared__ __attribute__((aligned(16))) float dzr[ROWS * WS];   // dz [row][12] (columns 10, 11 hold 0)
    __shared__ __attribute__((aligned(16))) float pan[16 * PS];     // dx tiles: dz1 [16 rows][H], stride H + 4
    __shared__ float redm[8][8][64];                                // (registers 4 .. 7: the second column half of a wide dx tile)
    __shared__ float bsum[8][64];
    __shared__ double red[8][4];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int i16 = lane & 15, grp = lane >> 4;
    const int srow = t >> 2, sub = t & 3;
    const int mt = p.m;                                              // all rows of this call
    const int n_dw = q.tiles_in * TH;
    const int blk = (int)blockIdx.x - G;
    const bool is_dw = blk < n_dw;
    const bool wide = q.dx_wide != 0;                                // block-uniform: dx tiles of 16 rows x 32 inputs
    const int n_in = q.n_in;
    // tile coordinates.  dW: tm over the inputs, tn over the hidden units; dx: tm over the rows, tn over the inputs
    int tm, tn;
    if (is_dw) {
        if (q.xcd && q.tiles_in % 2 == 0) {          // XCD-aware order: XCD x takes inputs' half x % 2, units' quarter x / 2
            const int xcd = blk & 7, idx = blk >> 3, pm = q.tiles_in / 2;
            tm = (xcd & 1) * pm + idx % pm;
            tn = (xcd >> 1) * (TH / 4) + idx / pm;
        } else { tm = blk % q.tiles_in; tn = blk / q.tiles_in; }
    } else {
        const int b2 = blk - n_dw, tr = (mt + 15) / 16;
        const int tcols = wide ? q.tiles_in / 2 : q.tiles_in;        // dx tile columns (16 or 32 inputs each)
        if (q.xcd && tr % 2 == 0 && tcols % 4 == 0 && n_dw % 8 == 0) {
            const int xcd = b2 & 7, idx = b2 >> 3, pm = tr / 2;
            tm = (xcd & 1) * pm + idx % pm;
            tn = (xcd >> 1) * (tcols / 4) + idx / pm;
        } else { tm = b2 % tr; tn = b2 / tr; }
    }
    const int m0 = tm * 16, n0 = tn * ((!is_dw && wide) ? 32 : 16);
    // SH == 2 (statistics from memory): any number of rows in blocks of 128.  A dW tile contracts over ALL rows — it walks the
    // blocks and keeps accumulating; a dx tile lives in ONE block (its 16 rows) and derives only that block's dz
    const int nb = RB ? (mt + ROWS - 1) / ROWS : 1;
    const int rb_lo = (RB && !is_dw) ? m0 / ROWS : 0, rb_hi = is_dw ? nb : rb_lo + 1;
    const int m0l = is_dw ? m0 : m0 - rb_lo * ROWS;                  // dx: first tile row inside its block

    // ---- the loads that do not depend on the row block, up front.  The dz1 slice of a tile is itself a 16x16x4 MFMA product
    //   P[row][unit] = sum_c dz[row][c] W2[unit][c]    (K = 12: 10 classes + 2 zero columns, 3 MFMAs per wave)
    // dW tile: wave w takes rows 16 w .. 16 w + 15 x units n0 .. n0 + 15; dx tile: rows m0 .. m0 + 15 x units 16 w .. 16 w + 15.
    // Lane (i16, grp) supplies B[k = grp][n = i16] = W2[unit i16][class 4 s + grp] and receives P[row 4 grp + r][unit i16],
    // r = 0..3 — for the dW tile that IS the B fragment of the tile product (b[j] = dz1[16 w + 4 grp + j][n0 + i16]).
    const int urow = (is_dw ? n0 : 16 * wid) + i16;                  // this lane's hidden unit
    const int prow0 = (is_dw ? 16 * wid : m0l) + 4 * grp;            // first of this lane's 4 panel rows (inside the block)
    float w2f[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) w2f[s3] = p.w[(size_t)urow * C + min(4 * s3 + grp, C - 1)];
    f32x4 bf = {0.f, 0.f, 0.f, 0.f}, bf1 = {0.f, 0.f, 0.f, 0.f};     // dx: W1 fragment(s)
    float e_pre = 0.f, e_pre1 = 0.f;             // dx: mask source(s)
    const int e_r = (t >> 6) & 3, e_ln = t & 63;
    if (!is_dw) {
        bf = *reinterpret_cast<const f32x4*>(q.w1 + (size_t)(n0 + i16) * H + 16 * wid + 4 * grp);
        if (wide) bf1 = *reinterpret_cast<const f32x4*>(q.w1 + (size_t)(n0 + 16 + i16) * H + 16 * wid + 4 * grp);
        if (t < 256) {
            const float* xr = q.x + (size_t)min(m0 + (e_ln >> 4) * 4 + e_r, mt - 1) * n_in + n0 + (e_ln & 15);
            e_pre = xr[0];
            if (wide) e_pre1 = xr[16];
        }
    }
    if (grp >= 2) w2f[2] = 0.f;                  // classes 10, 11 do not exist (the clamped address read class 9)
    static_assert(C == 10, "the zero columns of the K = 12 product are written for 10 classes");

    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bs = 0.f;
    const f32x4 bias4 = head_stage_bias<C, NP>(p, t);
    float ext_ms[2] = {0.f, 0.f};                // SH >= 2: the batch's {max, sum-exp}, merged once for all row blocks
    if constexpr (SH >= 2) head_merge_pairs(p.ext_pairs, p.ext_n, lane, ext_ms[0], ext_ms[1]);
    if constexpr (SH == 3) {                     // the first tile workgroup is the launch's sender
        __shared__ float xm[2];
        tnn::p2p::xchg_merge<512>(p.xc, p.xw, ext_ms[0], ext_ms[1], blk == 0, xm);
    }
    // the global reads of one row block: staged logits / labels, the mask source rows of a1, (dW) the x fragment
    HeadStage<C, NP> stg_nx;
    float a1m_nx[4], af_nx[4] = {0.f, 0.f, 0.f, 0.f};
    auto request_block = [&](const int rb) {
        int row0 = rb * ROWS, tt = t;
        if constexpr (RB) { asm volatile("" : "+s"(row0)); asm volatile("" : "+v"(tt)); }   // see head_block: keeps the addresses out of loop-carried registers
        const int m = RB ? min(ROWS, mt - row0) : mt;
        head_stage_request_rb<C, NP>(p, tt, stg_nx, m, row0);
#pragma unroll
        for (int r = 0; r < 4; ++r) a1m_nx[r] = p.a[(size_t)(row0 + min(prow0 + r, m - 1)) * H + urow];
        if (is_dw) {
#pragma unroll
            for (int j = 0; j < 4; ++j) af_nx[j] = q.x[(size_t)(row0 + min(16 * wid + 4 * grp + j, m - 1)) * n_in + m0 + i16];
        }
    };
    constexpr bool PF = RB && true;
    if (PF) request_block(rb_lo);
    for (int rb = rb_lo; rb < rb_hi; ++rb) {
    if (!PF) request_block(rb);
    const int row0 = rb * ROWS;
    const int m = RB ? min(ROWS, mt - row0) : mt;                    // rows of this block
    const bool slive = srow < m;
    const int sr = min(srow, m - 1);
    if (rb > rb_lo) __syncthreads();             // the staged logits of the previous block have been read by every wave
    float a1m[4], af[4];                         // af: the dW tile's x fragment
#pragma unroll
    for (int r = 0; r < 4; ++r) { a1m[r] = a1m_nx[r]; af[r] = af_nx[r]; }
    head_stage_store_rb<C, NP>(p, t, stg_nx, zs, ys, m, row0, bias4);
    // the staged values are in LDS, their registers are free: the next block's reads travel while this one is worked on
    if (PF && rb + 1 < rb_hi) request_block(rb + 1);
    __syncthreads();
    float zc[3], yc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        zc[i] = zs[sr * C + min(sub + 4 * i, C - 1)];
        yc[i] = ys[sr * C + min(sub + 4 * i, C - 1)];
    }
    if constexpr (CUT == 1) {
        q.dx[(size_t)(blk % 64) * 512 + t] = zc[0] + zc[1] + zc[2] + yc[0] + w2f[0] + w2f[1] + w2f[2] + a1m[0] + a1m[1] + a1m[2] + a1m[3] + af[0] + af[1] + af[2] + af[3] + bf[0] + e_pre;
        return;
    }
    HeadStats st;
    if constexpr (SH >= 2) head_stats<C, false, true>(zc, yc, slive, sub, lane, wid, red, st, false, ext_ms, 0);
    else head_stats<C, false>(zc, yc, slive, sub, lane, wid, red, st);
    if constexpr (CUT == 2) {
        q.dx[(size_t)(blk % 64) * 512 + t] = (float)st.S + st.M + st.ec[0] + st.eyc[1] + w2f[0] + w2f[1] + w2f[2] + a1m[0] + a1m[1] + a1m[2] + a1m[3] + af[0] + af[1] + af[2] + af[3] + bf[0] + e_pre;
        return;
    }
    float m_norm = (float)m;
    if constexpr (SH != 0) m_norm = (float)p.m_global;
    {
        const float sf = slive ? expf(st.mx - st.M) * __builtin_amdgcn_rcpf((float)st.S) : 0.f;
        const float uf = slive ? __builtin_amdgcn_rcpf(m_norm * st.urow) : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (st.valid[i]) dzr[srow * WS + sub + 4 * i] = st.ec[i] * sf - st.eyc[i] * uf;      // 0 in the padding rows
        if (sub >= 2) dzr[srow * WS + 8 + sub] = 0.f;
    }
    // dW tile: wave w reads back only the 16 rows it wrote itself (srow = 16 w + lane / 4) — no workgroup barrier, the
    // LDS queue of a wave is in order; dx tile: the 16 rows of the tile were written by wave m0l / 16
    if (is_dw) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
    if constexpr (CUT == 3) {
        q.dx[(size_t)(blk % 64) * 512 + t] = dzr[t] + w2f[0] + w2f[1] + w2f[2] + a1m[0] + a1m[1] + a1m[2] + a1m[3] + af[0] + af[1] + af[2] + af[3] + bf[0] + e_pre;
        return;
    }

    f32x4 pz = {0.f, 0.f, 0.f, 0.f};
    {
        const float* drow = dzr + ((is_dw ? 16 * wid : m0l) + i16) * WS + grp;         // A[m = i16][k = grp] = dz[row][4 s + grp]
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) pz = __builtin_amdgcn_mfma_f32_16x16x4f32(drow[4 * s3], w2f[s3], pz, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) pz[r] = (__float_as_uint(a1m[r]) >> 31) ? 0.f : pz[r];   // rows >= m: dz = 0 there
    }
    if (is_dw) {
        if constexpr (CUT == 4) {
            q.dx[(size_t)(blk % 64) * 512 + t] = pz[0] + pz[1] + pz[2] + pz[3] + af[0] + af[1] + af[2] + af[3];
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], pz[j], acc, 0, 0, 0);
        bs += (pz[0] + pz[1]) + (pz[2] + pz[3]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) pan[(4 * grp + r) * PS + 16 * wid + i16] = pz[r];
        __syncthreads();
        if constexpr (CUT == 4) {
            q.dx[(size_t)(blk % 64) * 512 + t] = pan[t] + bf[0] + e_pre;
            return;
        }
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(pan + i16 * PS + 16 * wid + 4 * grp);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bf[j], acc, 0, 0, 0);
        if (wide) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bf1[j], acc1, 0, 0, 0);
        }
    }
    }   // row blocks
#pragma unroll
    for (int r = 0; r < 4; ++r) redm[wid][r][lane] = acc[r];
    if (!is_dw && wide) {
#pragma unroll
        for (int r = 0; r < 4; ++r) redm[wid][4 + r][lane] = acc1[r];
    }
    bsum[wid][lane] = bs;
    __syncthreads();
    if (t < 256) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += redm[w][e_r][e_ln];
        const int row = m0 + (e_ln >> 4) * 4 + e_r, col = n0 + (e_ln & 15);           // 16x16x4 C/D layout
        if (is_dw) q.dw1[(size_t)row * H + col] = s;
        else if (row < mt) {
            q.dx[(size_t)row * n_in + col] = (__float_as_uint(e_pre) >> 31) ? 0.f : s;
            if (wide) {
                float s1 = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) s1 += redm[w][4 + e_r][e_ln];
                q.dx[(size_t)row * n_in + col + 16] = (__float_as_uint(e_pre1) >> 31) ? 0.f : s1;
            }
        }
    }
    if (is_dw && tm == 0 && t < 16) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += (bsum[w][t] + bsum[w][16 + t]) + (bsum[w][32 + t] + bsum[w][48 + t]);
        q.db1[n0 + t] = s;
    }
    TNN_STEP_STAMP_ACKED(g_step_trace_head, 0, 3);
}

// ------------------------------------------------------------------------------------------------------------------
// GENERIC merged head + hidden-layer backward: the same launch as mlp_head_bwd_kernel for any classifier head with
// H % 16 == 0, 16 <= H <= 256 hidden units, C <= 16 classes, n_in % 16 == 0 inputs of the hidden layer — e.g. the
// 80 -> 32 -> 10 tail of the reference's own example net (examples/mnist/run.py:59-69, hidden widths padded to multiples of
// 16 by the trainer).  Plain VALU loops over LDS panels, 256 threads, no shape baked in: it exists so that the 2L - 2 launch
// step is not tied to the benchmark's 128 -> 10 head; the tuned kernels above keep that shape.
//   statistics INSIDE (ext_pairs == NULL, <= 128 rows): every workgroup reduces the whole batch's {M, S} itself;
//   statistics from MEMORY (ext_pairs: ext_n {M_q, S_q} pairs left by the forward launch's tail, tnn_dense_fwd_head_partials_stats
//   [+ an all-gather over the ranks]; <= 1024 rows): nothing couples the rows inside the launch, so the head and dW1 roles walk
//   them in blocks of 128 with their sums in registers, a dx tile only looks at its own 16 rows, and 1 / m is the GLOBAL batch's
//   (core/losses.py:26-32 — the data-parallel step and every batch of more than 128 rows).
//     every workgroup : logits = bias + sum of the H / 16 partial tiles, dz [rows][C] (LDS)
//     blocks [0, H/16)            : dW2 rows [16 g, 16 g + 16) = a^T dz; block 0 also db2, loss, stats, beta powers; the last one
//                                   writes logits / dz out
//     blocks [H/16, + n_dw)       : dW1 tile (16 inputs x 16 units) = x^T dz1[:, 16 units], db1 (tiles of the first tile row);
//                                   dz1 = (dz W2^T) * !signbit(a) is derived in LDS, never stored
//     the rest                    : dx tile (16 rows x 16 inputs) = (dz1[16 rows, :] W1^T) * !signbit(x)
struct HeadGenArgs {
    int m, H, C, n_in;
    int m_global, ext_n;             // ext_pairs != NULL: rows of the GLOBAL batch, number of pairs (<= 64)
    const float* ext_pairs;
    int xw;                          // size of that group (by value)
    const tnn::p2p::XchgCtx* xc;     // statistics inside + DEFERRED exchange (<= 128 rows per rank; tnn_p2p.h: XchgCtx): every workgroup
                                     // reduces the shard's pair, workgroup 0 pushes it to the peers, all merge the ranks' pairs
    const float *a, *w, *b, *y, *zpart, *x, *w1;
    float *logits, *dz, *stats, *loss, *dw, *db, *dw1, *db1, *dx;
    double* tick;
    double b1, b2;
    float* ws;                       // row blocks in parallel: (H / 16 + n_in / 16 * H / 16) x ceil(m / 128) slots of 288 floats ...
    unsigned* tickets;               // ... and one arrival counter per unit (zero between launches); NULL = walk the blocks
};

__global__ __launch_bounds__(256) void mlp_head_bwd_generic_kernel(HeadGenArgs p) {
    constexpr int ROWS = 128, CS = 17, HMAX = 256, CMAX = 16;
    __shared__ float dzs[ROWS * CS];                 // dz [row of the block][class], stride 17
    __shared__ float pan[ROWS * CS];                 // head: a[:, 16 g ..] / dW1: dz1 [row][16 units] / dx: unused
    __shared__ float xs[ROWS * CS];                  // dW1: x[:, 16 inputs]
    __shared__ float w2s[HMAX * CMAX];               // W2 [H][C]
    __shared__ float pz[16 * (HMAX + 1)];            // dx: dz1 [16 rows][H]
    __shared__ float w1s[16 * (HMAX + 1)];           // dx: W1 [16 inputs][H]
    __shared__ float r_mx[ROWS], r_s[ROWS], r_lq[ROWS];
    __shared__ double red[4];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int m = p.m, H = p.H, C = p.C, n_in = p.n_in;
    const int G = H / 16, tiles_in = n_in / 16, n_dw = tiles_in * G, np = H / 16;
    const int blk = (int)blockIdx.x;
    const bool ext = p.ext_pairs != nullptr;         // block-uniform
    const bool dp = ext || p.xc != nullptr;          // data parallel: 1 / m is the GLOBAL batch's, the loss written is this rank's share
    __shared__ float xm[2];
    double pw0 = 0.0, pw1 = 0.0;
    if (blk == 0 && t == 0 && p.tick) { pw0 = p.tick[0]; pw1 = p.tick[1]; }
    float M = 0.f;
    double S = 1.0, L = 0.0;                         // L (statistics inside): sum over rows of log(e . y) + row max
    if (ext) {
        float Mx, Sx;
        head_merge_pairs(p.ext_pairs, p.ext_n, lane, Mx, Sx);      // every wave computes the same pair
        M = Mx;
        S = (double)Sx;
    }
    const double inv_m = 1.0 / (double)(dp ? p.m_global : m);
    // W2 -> LDS (every role needs it except the head's, which needs only dz): coalesced
    for (int i = t; i < H * C; i += 256) w2s[i] = p.w[i];

    // dz of rows [row0, row0 + n) into dzs[0 .. n) (rows >= m: zeros); returns this thread's row's log(e . y) + row max
    // (threads >= n, rows >= m: 0).  Statistics inside: n = 128 = the whole batch, M / S / L are reduced here.
    // Barriers inside; dzs is complete for every thread when it returns.
    auto block_dz = [&](const int row0, const int n, const bool write_out) -> float {
        float e[CMAX], yv[CMAX];
        float mx = -INFINITY, srow = 0.f, qy = 1.f;
        const int r = row0 + t;
        const bool live = t < n && r < m;
        if (live) {
            // every load of a partial tile is requested before the first one is used (tile loop outside, the row's classes
            // unrolled inside: np round trips to L2 per row instead of np x C dependent ones — 13.4 -> ~8 us per launch at 128 rows)
#pragma unroll
            for (int k = 0; k < CMAX; ++k) {
                e[k] = k < C ? p.b[k] : 0.f;
                yv[k] = k < C ? p.y[(size_t)r * C + k] : 0.f;
            }
            for (int tn = 0; tn < np; ++tn) {
                const float* zp = p.zpart + ((size_t)tn * m + r) * C;
                float part[CMAX];
#pragma unroll
                for (int k = 0; k < CMAX; ++k) part[k] = k < C ? zp[k] : 0.f;
#pragma unroll
                for (int k = 0; k < CMAX; ++k) e[k] += part[k];                 // tile order: the same sum as before
            }
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < C) mx = fmaxf(mx, e[k]);          // e[k]: the logit for now
            if (write_out && p.logits) {
#pragma unroll
                for (int k = 0; k < CMAX; ++k) if (k < C) p.logits[(size_t)r * C + k] = e[k];
            }
            qy = 0.f;
#pragma unroll
            for (int k = 0; k < CMAX; ++k) {
                e[k] = k < C ? expf(e[k] - mx) : 0.f;
                srow += e[k];
                qy += e[k] * yv[k];
            }
        }
        if (!ext) {
            if (t < ROWS) {
                r_mx[t] = live ? mx : -INFINITY;
                r_s[t] = live ? srow : 0.f;
                r_lq[t] = live ? mx + logf(qy) : 0.f;
            }
            __syncthreads();
            if (wid == 0) {
                float M0 = fmaxf(r_mx[lane], r_mx[lane + 64]);
                M0 = tnn::wave_max_dpp(M0);                   // (DPP trees: ~100 cycles each where a 64-bit __shfl_xor tree costs ~700)
                double S0 = (double)r_s[lane] * (double)expf(r_mx[lane] - M0) + (double)r_s[lane + 64] * (double)expf(r_mx[lane + 64] - M0);
                double L0 = (double)r_lq[lane] + (double)r_lq[lane + 64];
                S0 = tnn::wave_sum_dpp(S0);
                L0 = tnn::wave_sum_dpp(L0);
                if (lane == 0) { red[0] = (double)M0; red[1] = S0; red[2] = L0; }
            }
            __syncthreads();
            M = (float)red[0];
            S = red[1];
            L = red[2];
            if (p.xc != nullptr) {                        // block-uniform: the shard's pair -> the batch's (all ranks)
                float Sx = (float)S;
                tnn::p2p::xchg_merge<256>(p.xc, p.xw, M, Sx, blk == 0, xm);
                S = (double)Sx;
            }
        }
        if (t < n) {
            const float sf = live ? expf(mx - M) / (float)S : 0.f, uf = live ? (float)inv_m / qy : 0.f;
#pragma unroll
            for (int k = 0; k < CMAX; ++k) {
                const float d = live ? e[k] * sf - e[k] * yv[k] * uf : 0.f;
                dzs[t * CS + k] = d;                      // classes >= C: 0
                if (write_out && p.dz && live && k < C) p.dz[(size_t)r * C + k] = d;
            }
        }
        __syncthreads();
        return live ? mx + logf(qy) : 0.f;
    };
    const int nblk = ext ? (m + ROWS - 1) / ROWS : 1;
    // Row blocks in PARALLEL (p.ws given, more than one block): unit u (a dW2 row group or a dW1 tile) is worked on by nblk
    // workgroups, one per 128-row block; each leaves its partial sums in a slot of the workspace (system-scope stores) and draws
    // a ticket of the unit's arrival counter; the LAST one adds the nblk partials in BLOCK ORDER (the same sum whichever block
    // finishes last) and writes the result.  Without it (the <= 128-row step, or no workspace) a unit's one workgroup walks
    // the blocks itself.  (1024 rows of the reference's own net: 84 us with 12 workgroups walking 8 blocks each.)
    const bool par = p.ws != nullptr && nblk > 1;           // block-uniform
    const int units = G + n_dw, role_blocks = par ? units * nblk : units;
    constexpr int SLOT = 288;                                // floats per (unit, block): [0, 256) sums, [256, 272) bias sums, [272, 274) a double
    __shared__ int last_flag;
    // publish this workgroup's partials and find out whether it is the unit's last; `mine` = this thread's main sum (t < n_main),
    // `bsum` = its bias sum (t < n_b), lsum_total (thread 0) = the block's sum of log terms
    auto publish = [&](const int unit, const int rb, const float mine, const int n_main, const float bsum, const int n_b,
                       const double lsum_total) -> bool {
        float* slot = p.ws + ((size_t)unit * nblk + rb) * SLOT;
        if (t < n_main) __hip_atomic_store(slot + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (t < n_b) __hip_atomic_store(slot + 256 + t, bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (t == 0) __hip_atomic_store(reinterpret_cast<double*>(slot + 272), lsum_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            const unsigned prev = __hip_atomic_fetch_add(p.tickets + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == (unsigned)nblk - 1 ? 1 : 0;
            if (last) __hip_atomic_store(p.tickets + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // graph replays start from 0
            last_flag = last;
        }
        __syncthreads();
        return last_flag != 0;
    };
    auto gather = [&](const int unit, const int off) -> float {      // the nblk partials of element `off`, added in block order
        float acc = 0.f;
        for (int rb = 0; rb < nblk; ++rb)
            acc += __hip_atomic_load(p.ws + ((size_t)unit * nblk + rb) * SLOT + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return acc;
    };

    if (blk < role_blocks && (par ? blk / nblk : blk) < G) {
        // ---- head role: dW2 rows [16 g, 16 g + 16)
        const int g = par ? blk / nblk : blk, rb_lo = par ? blk % nblk : 0, rb_hi = par ? rb_lo + 1 : nblk;
        const int j = t / C, c = t - j * C;               // t < 16 C: this thread's element
        float s0 = 0.f, s1 = 0.f, dbs = 0.f;
        double lsum = 0.0;
        for (int rb = rb_lo; rb < rb_hi; ++rb) {
            const int row0 = rb * ROWS;
            const float lq = block_dz(row0, ROWS, g == G - 1);
            if (ext && g == 0) lsum += (double)lq;
#pragma unroll
            for (int i = t; i < ROWS * 16; i += 256) {      // (8 trips, unrolled: the 8 loads are requested together)
                const int r = i >> 4, jj = i & 15;
                pan[r * CS + jj] = row0 + r < m ? p.a[(size_t)(row0 + r) * H + 16 * g + jj] : 0.f;
            }
            __syncthreads();
            if (t < 16 * C) {
#pragma unroll 8
                for (int r = 0; r < ROWS; r += 2) {
                    s0 = fmaf(pan[r * CS + j], dzs[r * CS + c], s0);          // |a|: the sign bit is the ReLU mask, a >= 0 in value
                    s1 = fmaf(pan[(r + 1) * CS + j], dzs[(r + 1) * CS + c], s1);
                }
            }
            if (g == 0 && t < C)
                for (int r = 0; r < ROWS; ++r) dbs += dzs[r * CS + t];
            if (rb + 1 < rb_hi) __syncthreads();          // dzs / pan are rewritten by the next block
        }
        if (g == 0 && ext) {                              // sum of the rows' log terms: DPP-free, four waves through LDS
            lsum = tnn::wave_sum(lsum);
            __syncthreads();
            if (lane == 0) red[wid] = lsum;
            __syncthreads();
            L = (red[0] + red[1]) + (red[2] + red[3]);
        }
        float dw_v = s0 + s1;
        if (par) {
            if (!publish(g, rb_lo, dw_v, 16 * C, dbs, g == 0 ? C : 0, g == 0 ? L : 0.0)) return;
            if (t < 16 * C) dw_v = gather(g, t);
            if (g == 0) {
                if (t < C) dbs = gather(g, 256 + t);
                L = 0.0;
                for (int rb = 0; rb < nblk; ++rb)
                    L += __hip_atomic_load(reinterpret_cast<const double*>(p.ws + ((size_t)g * nblk + rb) * SLOT + 272), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (t < 16 * C) p.dw[(size_t)(16 * g + j) * C + c] = dw_v;
        if (g == 0) {
            if (t < C) p.db[t] = dbs;
            if (t == 0) {
                // data parallel: this rank's SHARE of the global loss (the all-reduce of the gradient arena sums the shares)
                if (p.loss) p.loss[0] = dp ? (float)((((double)logf((float)S) + (double)M) * (double)m - L) * inv_m)
                                           : (float)((double)logf((float)S) + (double)M - L * inv_m);
                if (p.stats) { p.stats[0] = M; p.stats[1] = (float)S; }
                if (p.tick) {
                    if (par) { pw0 = p.tick[0]; pw1 = p.tick[1]; }       // (this workgroup is not necessarily block 0)
                    p.tick[0] = pw0 * p.b1;
                    p.tick[1] = pw1 * p.b2;
                }
            }
        }
        return;
    }
    if (blk < role_blocks) {
        // ---- dW1 tile: inputs [m0, m0 + 16) x units [n0, n0 + 16)
        const int unit = par ? blk / nblk : blk, rb_lo = par ? blk % nblk : 0, rb_hi = par ? rb_lo + 1 : nblk;
        const int b2 = unit - G, tm = b2 % tiles_in, tn = b2 / tiles_in, m0 = tm * 16, n0 = tn * 16;
        const int ti = t >> 4, tj = t & 15;
        float s0 = 0.f, s1 = 0.f, dbs = 0.f;
        for (int rb = rb_lo; rb < rb_hi; ++rb) {
            const int row0 = rb * ROWS;
            block_dz(row0, ROWS, false);
            // (8 trips: the 16 global loads are requested together, then the dz1 elements are computed)
            float av8[8], xv8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = t + 256 * u, r = i >> 4, j = i & 15;
                const bool ok = row0 + r < m;
                av8[u] = ok ? p.a[(size_t)(row0 + r) * H + n0 + j] : 0.f;
                xv8[u] = ok ? p.x[(size_t)(row0 + r) * n_in + m0 + j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = t + 256 * u, r = i >> 4, j = i & 15;
                float acc = 0.f;
                for (int c = 0; c < C; ++c) acc = fmaf(dzs[r * CS + c], w2s[(n0 + j) * C + c], acc);
                pan[r * CS + j] = (row0 + r < m && !(__float_as_uint(av8[u]) >> 31)) ? acc : 0.f;
                xs[r * CS + j] = xv8[u];
            }
            __syncthreads();
#pragma unroll 8
            for (int r = 0; r < ROWS; r += 2) {
                s0 = fmaf(xs[r * CS + ti], pan[r * CS + tj], s0);
                s1 = fmaf(xs[(r + 1) * CS + ti], pan[(r + 1) * CS + tj], s1);
            }
            if (tm == 0 && t < 16)
                for (int r = 0; r < ROWS; ++r) dbs += pan[r * CS + t];
            if (rb + 1 < rb_hi) __syncthreads();
        }
        float dw_v = s0 + s1;
        if (par) {
            if (!publish(unit, rb_lo, dw_v, 256, dbs, tm == 0 ? 16 : 0, 0.0)) return;
            dw_v = gather(unit, t);
            if (tm == 0 && t < 16) dbs = gather(unit, 256 + t);
        }
        p.dw1[(size_t)(m0 + ti) * H + n0 + tj] = dw_v;
        if (tm == 0 && t < 16) p.db1[n0 + t] = dbs;
        return;
    }
    // ---- dx tile: rows [m0, m0 + 16) x inputs [n0, n0 + 16)
    {
        const int b3 = blk - role_blocks, tr = (m + 15) / 16, tm = b3 % tr, tn = b3 / tr, m0 = tm * 16, n0 = tn * 16;
        const int HS = H + 1;
        // statistics from memory: dz of this tile's 16 rows only; inside: the whole (<= 128-row) batch is needed for M and S
        const int dz0 = ext ? m0 : 0;
        block_dz(dz0, ext ? 16 : ROWS, false);
        for (int i = t; i < 16 * H; i += 256) {
            const int rr = i / H, j = i - rr * H, r = m0 + rr;
            float d1 = 0.f;
            if (r < m) {
                const float av = p.a[(size_t)r * H + j];
                float acc = 0.f;
                for (int c = 0; c < C; ++c) acc = fmaf(dzs[(r - dz0) * CS + c], w2s[j * C + c], acc);
                d1 = (__float_as_uint(av) >> 31) ? 0.f : acc;
            }
            pz[rr * HS + j] = d1;
            w1s[rr * HS + j] = p.w1[(size_t)(n0 + rr) * H + j];           // row rr of this tile's 16 W1 rows
        }
        __syncthreads();
        const int rr = t >> 4, i = t & 15, r = m0 + rr;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int j = 0; j < H; j += 2) {
            s0 = fmaf(pz[rr * HS + j], w1s[i * HS + j], s0);
            s1 = fmaf(pz[rr * HS + j + 1], w1s[i * HS + j + 1], s1);
        }
        if (r < m) {
            const float xv = p.x[(size_t)r * n_in + n0 + i];
            p.dx[(size_t)r * n_in + n0 + i] = (__float_as_uint(xv) >> 31) ? 0.f : s0 + s1;
        }
    }
}

// Workspace of the generic kernel's parallel row blocks: slots + arrival counters, one set per stream, allocated ONCE at its
// upper bound (the 16 MB cap; wider hidden layers walk the blocks serially) outside a capture — tnn_mlp_head_bwd_reserve at
// trainer creation, or the first eager call — and never returned or regrown: the pointers are kernel arguments of every
// hipGraph captured since (growing by hipFree + hipMalloc left earlier captures replaying into freed memory).  The counters
// are zeroed once — every launch leaves them at zero.
struct HeadWs {
    float* slots = nullptr;
    unsigned* tickets = nullptr;
    size_t slot_floats = 0;
    int units = 0;
};
std::mutex g_head_ws_mu;
std::unordered_map<hipStream_t, HeadWs> g_head_ws;

bool head_workspace(hipStream_t st, int units, int nblk, HeadWs* out) {
    std::lock_guard<std::mutex> lk(g_head_ws_mu);
    HeadWs& w = g_head_ws[st];
    constexpr size_t CAP_FLOATS = ((size_t)16 << 20) / 4;
    constexpr int MAX_UNITS = (int)(CAP_FLOATS / (2 * 288));       // a request has at least two row blocks
    const size_t need = (size_t)units * nblk * 288;
    if (need > CAP_FLOATS || units > MAX_UNITS) return false;
    if (w.slots != nullptr) { *out = w; return true; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    if (hipMalloc(&w.slots, CAP_FLOATS * 4) != hipSuccess) { (void)hipGetLastError(); w = HeadWs(); return false; }
    if (hipMalloc(&w.tickets, (size_t)MAX_UNITS * 4) != hipSuccess || hipMemset(w.tickets, 0, (size_t)MAX_UNITS * 4) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(w.slots);
        w = HeadWs();
        return false;
    }
    w.slot_floats = CAP_FLOATS;
    w.units = MAX_UNITS;
    *out = w;
    return true;
}

// ext: the statistics come from memory (row blocks: up to 1024 rows); else the workgroups reduce them (<= 128 rows)
bool head_bwd_generic_fits(int64_t rows, int64_t n_in, int64_t n_hidden, int64_t n_classes, int dtype, bool ext = false) {
    return dtype == TNN_F32 && rows >= 1 && rows <= (ext ? 1024 : 128) && n_hidden % 16 == 0 && n_hidden >= 16 &&
           n_hidden <= 256 && n_classes >= 1 && n_classes <= 16 && n_in >= 16 && n_in % 16 == 0;
}

bool head_multi_fits(int64_t rows, int64_t n_hidden, int64_t n_classes, int dtype) {
    return dtype == TNN_F32 && n_classes == 10 && n_hidden == 128 && rows >= 1 && rows <= 128;
}

int head_bwd_launch(const char* fn, const float* ext_pairs, int ext_n, bool whole_logits, int64_t m_global, int64_t rows, int64_t n_in, int64_t n_hidden, int64_t n_classes, const void* x, const void* w1,
                          const tnn::p2p::XchgCtx* xc, const int xw,
                          const void* a, const void* w, const void* b, const void* y, const void* logit_partials,
                          void* logits, void* dz, void* stats, void* loss, void* dw, void* db, void* dw1, void* db1,
                          void* dx, int dtype, void* adam_pows_f64, double b1, double b2) {
    TNN_NEED_INIT();
    TNN_REQUIRE(rows > 0 && n_in > 0 && n_hidden > 0 && n_classes > 0, "%s: empty head", fn);
    TNN_REQUIRE(x && w1 && a && w && b && y && logit_partials && dw && db && dw1 && db1 && dx,
                "%s: x, w1, a, w, b, y, logit_partials, dw, db, dw1, db1 and dx are required", fn);
    auto al = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    if (!head_multi_fits(1, n_hidden, n_classes, dtype) && !whole_logits &&
        head_bwd_generic_fits(rows, n_in, n_hidden, n_classes, dtype, ext_pairs != nullptr)) {
        // any other head the merged launch can take: the generic kernel (shapes as run-time arguments)
        TNN_REQUIRE(ext_pairs == nullptr || (m_global >= rows && ext_n >= 1 && ext_n <= 64), "%s: m_global < rows or bad pair count", fn);
        HeadGenArgs ga;
        ga.m = (int)rows; ga.H = (int)n_hidden; ga.C = (int)n_classes; ga.n_in = (int)n_in;
        ga.m_global = (int)m_global; ga.ext_n = ext_n; ga.ext_pairs = ext_pairs; ga.xc = xc; ga.xw = xw;
        TNN_REQUIRE(xc == nullptr || (ext_pairs == nullptr && m_global >= rows),
                    "%s: the deferred exchange of a generic head is the <= 128-row form (statistics inside)", fn);
        ga.a = (const float*)a; ga.w = (const float*)w; ga.b = (const float*)b; ga.y = (const float*)y;
        ga.zpart = (const float*)logit_partials; ga.x = (const float*)x; ga.w1 = (const float*)w1;
        ga.logits = (float*)logits; ga.dz = (float*)dz; ga.stats = (float*)stats; ga.loss = (float*)loss;
        ga.dw = (float*)dw; ga.db = (float*)db; ga.dw1 = (float*)dw1; ga.db1 = (float*)db1; ga.dx = (float*)dx;
        ga.tick = (double*)adam_pows_f64; ga.b1 = b1; ga.b2 = b2;
        const int gh = (int)(n_hidden / 16), ti = (int)(n_in / 16);
        const int units = gh + ti * gh, nblk = ext_pairs != nullptr ? (int)((rows + 127) / 128) : 1;
        ga.ws = nullptr; ga.tickets = nullptr;
        HeadWs hw;
        if (nblk > 1 && head_workspace(tnn::stream(), units, nblk, &hw)) { ga.ws = hw.slots; ga.tickets = hw.tickets; }
        const int grid = units * (ga.ws ? nblk : 1) + (int)((rows + 15) / 16) * ti;
        hipLaunchKernelGGL(mlp_head_bwd_generic_kernel, grid, 256, 0, tnn::stream(), ga);
        TNN_LAUNCH_OK();
        return 0;
    }
    // rows: <= 128 when the workgroups reduce the statistics themselves; with the statistics taken from memory (ext_pairs)
    // nothing couples the rows inside the launch and they are walked in blocks of 128 (<= 1024: 8 blocks)
    TNN_REQUIRE(head_multi_fits(ext_pairs ? (rows <= 1024 ? 1 : rows) : rows, n_hidden, n_classes, dtype) && n_in % 16 == 0 &&
                    al(a) && al(w) && al(w1),
                "%s: this head does not fit the merged form (tnn_mlp_head_fits, n_in %% 16 == 0, 16-B aligned a / w / w1)", fn);
    HeadMArgs p;
    p.m = (int)rows;
    p.rpb = (int)((rows + 15) / 16);
    p.a = (const float*)a; p.w = (const float*)w; p.b = (const float*)b; p.y = (const float*)y;
    p.zpart = (const float*)logit_partials;
    p.vec = (rows % 2 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(logit_partials)) & 15) == 0) ? 1 : 0;
    p.m_global = (int)m_global;
    p.ext_pairs = ext_pairs; p.ext_n = ext_n; p.xc = xc; p.xw = xw;
    TNN_REQUIRE(!whole_logits || rows > 128, "%s: whole logits (n_pairs < 0) come from the row-panel forward, i.e. with more than 128 rows", fn);
    p.nparts = whole_logits ? 1 : 8;
    p.logits = (float*)logits; p.dz = (float*)dz; p.stats = (float*)stats; p.loss = (float*)loss;
    p.dw = (float*)dw; p.db = (float*)db; p.da = nullptr;
    p.tick = (double*)adam_pows_f64; p.b1 = b1; p.b2 = b2;
    HeadBwdArgs q;
    q.x = (const float*)x; q.w1 = (const float*)w1;
    q.dw1 = (float*)dw1; q.db1 = (float*)db1; q.dx = (float*)dx;
    q.n_in = (int)n_in; q.tiles_in = (int)(n_in / 16);
    q.xcd = 1;
    // dx tiles of 32 inputs (HeadBwdArgs::dx_wide; TNN_HEAD_DX_WIDE=0 / TNN_HEAD_DX_WIDE_RB=0: the 16-wide tiles at every row count / above 128 rows, A/B)
    static const bool dx_wide_off = getenv("TNN_HEAD_DX_WIDE") && atoi(getenv("TNN_HEAD_DX_WIDE")) == 0;
    static const bool dx_wide_rb_off = getenv("TNN_HEAD_DX_WIDE_RB") && atoi(getenv("TNN_HEAD_DX_WIDE_RB")) == 0;
    q.dx_wide = (q.tiles_in % 2 == 0 && !dx_wide_off && (rows <= 128 || !dx_wide_rb_off)) ? 1 : 0;
    const int grid = 16 + q.tiles_in * 8 + (int)((rows + 15) / 16) * (q.dx_wide ? q.tiles_in / 2 : q.tiles_in);
    if (xc != nullptr) {
        // data parallel, DEFERRED exchange: <= 128 rows — statistics inside as on one GPU, then exchanged; more — the panels' pairs
        // of THIS shard from memory (row-panel forward), merged, then exchanged
        TNN_REQUIRE(m_global >= rows && (rows <= 128 ? ext_pairs == nullptr : (ext_pairs != nullptr && ext_n >= 1 && ext_n <= 64)),
                    "%s: m_global < rows, or the pairs do not match the row count", fn);
        if (rows > 128) hipLaunchKernelGGL((mlp_head_bwd_rb_kernel<128, 10, 0, 3, true>), grid, 512, 0, tnn::stream(), p, q);
        else hipLaunchKernelGGL((mlp_head_bwd_kernel<128, 10, 0, 3>), grid, 512, 0, tnn::stream(), p, q);
        TNN_LAUNCH_OK();
        return 0;
    }
    if (ext_pairs != nullptr) {          // data parallel: the statistics come from the tail of the previous launch [+ all-gather]
        TNN_REQUIRE(m_global >= rows && ext_n >= 1 && ext_n <= 64, "%s: m_global < rows or bad pair count", fn);
        if (rows > 128) hipLaunchKernelGGL((mlp_head_bwd_rb_kernel<128, 10, 0, 2, true>), grid, 512, 0, tnn::stream(), p, q);
        else hipLaunchKernelGGL((mlp_head_bwd_kernel<128, 10, 0, 2>), grid, 512, 0, tnn::stream(), p, q);
        TNN_LAUNCH_OK();
        return 0;
    }
    hipLaunchKernelGGL((mlp_head_bwd_kernel<128, 10>), grid, 512, 0, tnn::stream(), p, q);
    TNN_LAUNCH_OK();
    return 0;
}


}  // namespace

extern "C" {

#ifdef TNN_STEP_TRACE
__attribute__((visibility("default"))) int tnn_debug_step_trace_head(unsigned long long* out, int n) {
    TNN_CHECK_HIP(hipDeviceSynchronize());
    TNN_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_trace_head), (size_t)(n < 4096 ? n : 4096) * 8));
    return 0;
}
#endif

int tnn_mlp_head(int64_t rows, int64_t n_hidden, int64_t n_classes, const void* a, const void* w,
                 const void* b, const void* y, void* logits, void* dz, void* stats, void* loss, void* dw,
                 void* db, void* da, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(rows > 0 && n_hidden > 0 && n_classes > 0, "tnn_mlp_head: empty head");
    TNN_REQUIRE(logits != nullptr && dz != nullptr && dw != nullptr && db != nullptr,
                "tnn_mlp_head: logits, dz, dw and db buffers are required");
    // the same maths as three launches (any shape / dtype)
    if (int rc = tnn_gemm_bias_act(0, 0, rows, n_classes, n_hidden, a, n_hidden, w, n_classes, b, TNN_ACT_NONE, 0,
                                   logits, n_classes, dtype))
        return rc;
    if (int rc = tnn_softmax_nll_fused(logits, y, rows, n_classes, stats, loss, dz, dtype)) return rc;
    return tnn_dense_bwd(rows, n_hidden, n_classes, a, dz, w, dw, db, da, a, dtype);
}

int tnn_mlp_head_fits(int64_t rows, int64_t n_hidden, int64_t n_classes, int dtype, int* fits) {
    TNN_REQUIRE(fits != nullptr, "tnn_mlp_head_fits: fits is NULL");
    *fits = head_multi_fits(rows, n_hidden, n_classes, dtype) ? 1 : 0;
    return 0;
}

int tnn_mlp_head_bwd_fits(int64_t rows, int64_t n_in, int64_t n_hidden, int64_t n_classes, int dtype, int* fits) {
    TNN_REQUIRE(fits != nullptr, "tnn_mlp_head_bwd_fits: fits is NULL");
    *fits = ((head_multi_fits(rows, n_hidden, n_classes, dtype) && n_in % 16 == 0 && n_in >= 16) ||
             head_bwd_generic_fits(rows, n_in, n_hidden, n_classes, dtype)) ? 1 : 0;
    return 0;
}

int tnn_mlp_head_bwd_reserve(int64_t max_rows, int64_t n_in, int64_t n_hidden, int64_t n_classes) {
    TNN_NEED_INIT();
    if (head_multi_fits(1, n_hidden, n_classes, TNN_F32) || !head_bwd_generic_fits(max_rows < 1024 ? max_rows : 1024, n_in, n_hidden, n_classes, TNN_F32, true))
        return 0;                                          // the tuned head, or a head the merged launch does not take
    const int nblk = (int)(((max_rows < 1024 ? max_rows : 1024) + 127) / 128);
    HeadWs hw;
    if (nblk > 1) (void)head_workspace(tnn::stream(), (int)(n_hidden / 16 + n_in / 16 * (n_hidden / 16)), nblk, &hw);   // (too large: serial form)
    return 0;
}

int tnn_mlp_head_tick(int64_t rows, int64_t n_hidden, int64_t n_classes, const void* a, const void* w,
                      const void* b, const void* y, const void* logit_partials, void* logits, void* dz, void* stats,
                      void* loss, void* dw, void* db, void* da, int dtype, void* adam_pows_f64, double b1, double b2) {
    TNN_NEED_INIT();
    TNN_REQUIRE(rows > 0 && n_hidden > 0 && n_classes > 0, "tnn_mlp_head_tick: empty head");
    TNN_REQUIRE(a && w && b && y && dw && db, "tnn_mlp_head_tick: a, w, b, y, dw and db are required");
    TNN_REQUIRE(head_multi_fits(rows, n_hidden, n_classes, dtype) && (reinterpret_cast<uintptr_t>(a) & 15) == 0,
                "tnn_mlp_head_tick: this head does not fit the one-launch form (ask tnn_mlp_head_fits first)");
    HeadMArgs p;
    p.m = (int)rows;
    p.rpb = (int)((rows + 15) / 16);
    p.a = (const float*)a; p.w = (const float*)w; p.b = (const float*)b; p.y = (const float*)y;
    p.zpart = (const float*)logit_partials;
    p.vec = (rows % 2 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(logit_partials)) & 15) == 0) ? 1 : 0;
    p.m_global = 0;
    p.ext_pairs = nullptr; p.ext_n = 0; p.xc = nullptr; p.xw = 1;
    p.nparts = 8;
    p.logits = (float*)logits; p.dz = (float*)dz; p.stats = (float*)stats; p.loss = (float*)loss;
    p.dw = (float*)dw; p.db = (float*)db; p.da = (float*)da;
    p.tick = (double*)adam_pows_f64; p.b1 = b1; p.b2 = b2;
    hipStream_t st = tnn::stream();
    if (p.zpart) hipLaunchKernelGGL((mlp_head_multi_kernel<128, 10, true>), 16, 512, 0, st, p);
    else hipLaunchKernelGGL((mlp_head_multi_kernel<128, 10, false>), 16, 512, 0, st, p);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_mlp_head_bwd_tick(int64_t rows, int64_t n_in, int64_t n_hidden, int64_t n_classes, const void* x, const void* w1,
                          const void* a, const void* w, const void* b, const void* y, const void* logit_partials,
                          void* logits, void* dz, void* stats, void* loss, void* dw, void* db, void* dw1, void* db1,
                          void* dx, int dtype, void* adam_pows_f64, double b1, double b2) {
    return head_bwd_launch("tnn_mlp_head_bwd_tick", nullptr, 0, false, 0, rows, n_in, n_hidden, n_classes, x, w1, nullptr, 1, a, w, b, y, logit_partials, logits,
                           dz, stats, loss, dw, db, dw1, db1, dx, dtype, adam_pows_f64, b1, b2);
}

int tnn_mlp_head_bwd_tick_ext(int64_t rows, int64_t m_global, int64_t n_in, int64_t n_hidden, int64_t n_classes,
                              const void* x, const void* w1, const void* a, const void* w, const void* b, const void* y,
                              const void* logit_partials, const void* stats_pairs, int n_pairs, void* logits, void* dz,
                              void* stats, void* loss, void* dw, void* db, void* dw1, void* db1, void* dx, int dtype,
                              void* adam_pows_f64, double b1, double b2) {
    TNN_REQUIRE(stats_pairs != nullptr && m_global >= 1, "tnn_mlp_head_bwd_tick_ext: stats_pairs and m_global are required");
    // n_pairs < 0: logit_partials holds WHOLE logits [rows][classes] without the bias (tnn_dense_fwd_rows_head_stats) and
    // there are -n_pairs pairs
    return head_bwd_launch("tnn_mlp_head_bwd_tick_ext", (const float*)stats_pairs, n_pairs < 0 ? -n_pairs : n_pairs, n_pairs < 0, m_global, rows, n_in, n_hidden,
                           n_classes, x, w1, nullptr, 1, a, w, b, y, logit_partials, logits, dz, stats, loss, dw, db, dw1, db1, dx, dtype,
                           adam_pows_f64, b1, b2);
}

// Does the merged head launch with the DEFERRED statistics exchange take this shape — and may it be launched here?  In that
// launch EVERY workgroup waits for the peers' pairs, so all ranks whose launches run on THIS GPU (the tests' shared-GPU groups;
// one in a one-process-per-GPU job) must fit the device together: ranks_on_my_device x grid <= resident workgroups of the
// kernel, which always leaves a lagging rank's earlier launches room to run.  TNN_DP_XCHG=2 skips that rule (experiments).
int tnn_mlp_head_bwd_xchg_fits(int64_t rows, int64_t n_in, int64_t n_hidden, int64_t n_classes, int dtype, int* fits) {
    TNN_REQUIRE(fits != nullptr, "tnn_mlp_head_bwd_xchg_fits: fits is NULL");
    *fits = 0;
    if (!tnn::initialised() || tnn::p2p_xchg_ctx() == nullptr || dtype != TNN_F32 || rows < 1 || n_in < 16 || n_in % 16) return 0;
    const bool tuned = head_multi_fits(1, n_hidden, n_classes, dtype);
    int grid = 0, per_cu = 0;
    if (tuned && rows <= 1024) {
        static const bool dx_wide_off = getenv("TNN_HEAD_DX_WIDE") && atoi(getenv("TNN_HEAD_DX_WIDE")) == 0;
        static const bool dx_wide_rb_off = getenv("TNN_HEAD_DX_WIDE_RB") && atoi(getenv("TNN_HEAD_DX_WIDE_RB")) == 0;
        const int ti = (int)(n_in / 16), dx_cols = (ti % 2 == 0 && !dx_wide_off && (rows <= 128 || !dx_wide_rb_off)) ? ti / 2 : ti;    // (head_bwd_launch's grid)
        grid = 16 + ti * 8 + (int)((rows + 15) / 16) * dx_cols;
        static int per_cu_small = -1, per_cu_rb = -1;
        if (per_cu_small < 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_small, mlp_head_bwd_kernel<128, 10, 0, 3>, 512, 0) != hipSuccess) per_cu_small = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_rb, mlp_head_bwd_rb_kernel<128, 10, 0, 3, true>, 512, 0) != hipSuccess) per_cu_rb = 0;
            (void)hipGetLastError();
        }
        per_cu = rows <= 128 ? per_cu_small : per_cu_rb;
    } else if (!tuned && head_bwd_generic_fits(rows, n_in, n_hidden, n_classes, dtype, false)) {
        const int gh = (int)(n_hidden / 16), ti = (int)(n_in / 16);
        grid = gh + ti * gh + (int)((rows + 15) / 16) * ti;
        static int per_cu_gen = -1;
        if (per_cu_gen < 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_gen, mlp_head_bwd_generic_kernel, 256, 0) != hipSuccess) per_cu_gen = 0;
            (void)hipGetLastError();
        }
        per_cu = per_cu_gen;
    } else {
        return 0;
    }
    const int shared = tnn::p2p_ranks_on_my_device();
    static const bool force = getenv("TNN_DP_XCHG") && atoi(getenv("TNN_DP_XCHG")) == 2;
    if (shared <= 1 || force || (int64_t)shared * grid <= (int64_t)per_cu * tnn::num_cus()) *fits = 1;
    return 0;
}

int tnn_mlp_head_bwd_tick_xchg(int64_t rows, int64_t m_global, int64_t n_in, int64_t n_hidden, int64_t n_classes,
                               const void* x, const void* w1, const void* a, const void* w, const void* b, const void* y,
                               const void* logit_partials, const void* shard_pairs, int n_pairs, void* logits, void* dz,
                               void* stats, void* loss, void* dw, void* db, void* dw1, void* db1, void* dx, int dtype,
                               void* adam_pows_f64, double b1, double b2) {
    // The merged head + hidden-backward launch of a data-parallel step on the peer-to-peer transport with the DEFERRED statistics
    // exchange (tnn_p2p.h: XchgCtx): the shard's {max, sum-exp} is reduced inside (n_pairs = 0: <= 128 rows, from the partial
    // logits) or merged from the row-panel forward's pairs (n_pairs < 0: -n_pairs pairs, logit_partials = whole logits), pushed
    // to the peers by one workgroup, and every workgroup merges the ranks' pairs.  The forward launch in front MUST have been
    // issued with exchange = 2 (it advances the sequence that tags the pairs).
    TNN_NEED_INIT();
    TNN_REQUIRE(m_global >= 1 && n_pairs <= 0 && (n_pairs == 0 || shard_pairs != nullptr),
                "tnn_mlp_head_bwd_tick_xchg: m_global >= 1, n_pairs = 0 (statistics inside) or < 0 (row-panel pairs of this shard)");
    if (int rc = tnn::p2p_refuse_if_failed("tnn_mlp_head_bwd_tick_xchg")) return rc;
    const tnn::p2p::XchgCtx* xc = tnn::p2p_xchg_ctx();
    TNN_REQUIRE(xc != nullptr, "tnn_mlp_head_bwd_tick_xchg: the peer-to-peer transport is not enabled");
    int rank = 0, world = 1;
    (void)tnn::p2p_world(&rank, &world);
    return head_bwd_launch("tnn_mlp_head_bwd_tick_xchg", n_pairs < 0 ? (const float*)shard_pairs : nullptr, -n_pairs, n_pairs < 0, m_global,
                           rows, n_in, n_hidden, n_classes, x, w1, xc, world, a, w, b, y, logit_partials, logits, dz, stats, loss, dw, db, dw1,
                           db1, dx, dtype, adam_pows_f64, b1, b2);
}

}  // extern "C"
