// Whole-batch softmax statistics of a classifier head (core/losses.py:24-32) shared by the head kernels (tnn_head.hip)
// and by the forward launch in front of them (tnn_gemm.hip: the LAST workgroup of the hidden layer's forward to finish
// reduces the shard's {max, sum-exp} — and on the xGMI peer-to-peer transport exchanges and merges them — so a
// data-parallel step needs neither a statistics launch nor workgroups that wait for a peer inside the head launch).
#pragma once
#include "tnn_internal.h"
#include "tnn_p2p.h"

namespace {

struct HeadStats {
    bool valid[3];
    float ec[3], eyc[3];             // exp(z - row max), times the label
    float mx, urow, M;               // row max, row sum of e * y, batch max
    double S, L;                     // batch sum-exp (relative to M), sum over rows of log u + row max
};

// Merge of ext_n {max, sum-exp} pairs from memory, one per lane (ext_n <= 64: the ranks' pairs, or one per 16-row panel of the
// row-panel forward — a serial loop of 64 dependent expf per call cost the 1024-row step 80 us): wave-wide DPP max, then the
// rescaled sum; every wave computes the same pair.
__device__ __forceinline__ void head_merge_pairs(const float* pairs, const int n, const int lane, float& M, float& S) {
    if (n == 1) {                                               // one already merged pair (uniform branch)
        M = pairs[0];
        S = pairs[1];
        return;
    }
    const bool has = lane < n;
    const float mq = has ? pairs[2 * lane] : -INFINITY, sq = has ? pairs[2 * lane + 1] : 0.f;
    M = tnn::wave_max_dpp(mq);
    S = tnn::wave_sum_dpp(has ? sq * expf(mq - M) : 0.f);
}

// zc / yc: this thread's three logits / labels of row (t >> 2) (classes sub, sub + 4, sub + 8); one barrier inside.
// LOSS = false (workgroups that only need dz): no sum of logs, cross-row sums in f32 — a 1280-term DPP tree is good to
// ~1e-6 relative, dz's tolerance is 1e-5 — which takes the f64 DPP reductions and the logf off their critical path.
// EXT: {M, S} of the batch come from memory — ext_n pairs {M_q, S_q} (one per rank of a data-parallel group, or one already
// merged pair) written by an EARLIER launch (head_tail_stats below [+ a collective]) and merged here; the workgroup then does no
// cross-row reduction at all, except the sum of logs in the one workgroup that writes the loss.
template <int C, bool LOSS = true, bool EXT = false>
__device__ __forceinline__ void head_stats(const float (&zc)[3], const float (&yc)[3], const bool slive, const int sub,
                                           const int lane, const int wid, double (*red)[4], HeadStats& o,
                                           const bool want_loss = true, const float* ext_pairs = nullptr, const int ext_n = 0) {
    // ---- whole-batch softmax statistics: FOUR threads per row (classes sub, sub + 4, sub + 8), row max / sums by
    // quad-permute DPP; then ONE combined reduction of {max, rescaled sum-exp, sum(log u + max)} over the workgroup
    // (DPP wave reductions + an 8-entry LDS exchange).  Row sums in f32 (10 terms), cross-row sums in f64 — the
    // arithmetic of nll_rows_body (tnn_nll_rows.h) up to summation order.
    bool (&valid)[3] = o.valid;
    float (&ec)[3] = o.ec, (&eyc)[3] = o.eyc;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        valid[i] = sub + 4 * i < C;
        if (valid[i]) mx = zc[i] > mx ? zc[i] : mx;
    }
    { float q = tnn::dpp_move<0xB1, 0xf>(-INFINITY, mx); mx = q > mx ? q : mx; }
    { float q = tnn::dpp_move<0x4E, 0xf>(-INFINITY, mx); mx = q > mx ? q : mx; }
    float srow_sum = 0.f, urow = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ec[i] = valid[i] ? expf(zc[i] - mx) : 0.f;
        eyc[i] = ec[i] * (valid[i] ? yc[i] : 0.f);
        srow_sum += ec[i];
        urow += eyc[i];
    }
    srow_sum += tnn::dpp_move<0xB1, 0xf>(0.f, srow_sum);
    srow_sum += tnn::dpp_move<0x4E, 0xf>(0.f, srow_sum);
    urow += tnn::dpp_move<0xB1, 0xf>(0.f, urow);
    urow += tnn::dpp_move<0x4E, 0xf>(0.f, urow);
    const bool counts = slive && sub == 0;                      // one lane per row feeds the cross-row sums
    if constexpr (EXT) {
        // the batch's pair: merged by the caller once per workgroup (head_merge_pairs) and handed over in ext_pairs[0..1] of
        // a two-float array in registers (ext_n is then 0), or merged here
        float Mx, Sx;
        if (ext_n == 0) { Mx = ext_pairs[0]; Sx = ext_pairs[1]; }
        else head_merge_pairs(ext_pairs, ext_n, lane, Mx, Sx);
        double Lx = 0.0;
        if (LOSS && want_loss) {                                // block-uniform: the loss-writing workgroup only
            const double wl = tnn::wave_sum_dpp(counts ? (double)logf(urow) + (double)mx : 0.0);
            if (lane == 0) red[wid][2] = wl;
            __syncthreads();
            Lx = red[lane & 7][2];
            Lx += tnn::dpp_move<0xB1, 0xf>(0.0, Lx);
            Lx += tnn::dpp_move<0x4E, 0xf>(0.0, Lx);
            Lx += tnn::dpp_move<0x141, 0xf>(0.0, Lx);
        }
        o.mx = mx; o.urow = urow; o.M = Mx; o.S = (double)Sx; o.L = Lx;
        return;
    }
    // A wave holds 16 rows: its {max, sum-exp relative to that max} in f32 (160 terms, DPP tree).  The eight waves' results
    // meet in LDS; every lane then takes entry (lane & 7) and an 8-lane DPP butterfly (quad_perm x 2, row_half_mirror)
    // leaves M, S (and L) in ALL lanes — one exp per lane instead of a serial 8-term loop per thread (measured: the
    // f64 loop + f64 64-lane reductions cost 1.5 us of the head's 4.4).  Across waves S and L are summed in f64 (LOSS).
    const float wm = tnn::wave_max_dpp(slive ? mx : -INFINITY);
    const float wsf = tnn::wave_sum_dpp(counts ? srow_sum * expf(mx - wm) : 0.f);
    const int w8 = lane & 7;
    if constexpr (!LOSS) {
        float* redf = reinterpret_cast<float*>(red);            // [8][2] floats in the same LDS words
        if (lane == 0) { redf[2 * wid] = wm; redf[2 * wid + 1] = wsf; }
        __syncthreads();
        const float rm = redf[2 * w8], rs = redf[2 * w8 + 1];
        float Mf = rm, q;
        q = tnn::dpp_move<0xB1, 0xf>(-INFINITY, Mf); Mf = q > Mf ? q : Mf;
        q = tnn::dpp_move<0x4E, 0xf>(-INFINITY, Mf); Mf = q > Mf ? q : Mf;
        q = tnn::dpp_move<0x141, 0xf>(-INFINITY, Mf); Mf = q > Mf ? q : Mf;
        float Sf = rm > -INFINITY ? rs * expf(rm - Mf) : 0.f;
        Sf += tnn::dpp_move<0xB1, 0xf>(0.f, Sf);
        Sf += tnn::dpp_move<0x4E, 0xf>(0.f, Sf);
        Sf += tnn::dpp_move<0x141, 0xf>(0.f, Sf);
        o.mx = mx; o.urow = urow; o.M = Mf; o.S = (double)Sf; o.L = 0.0;
        return;
    }
    double wlog = 0.0;
    if (want_loss) wlog = tnn::wave_sum_dpp(counts ? (double)logf(urow) + (double)mx : 0.0);     // block-uniform branch
    if (lane == 0) { red[wid][0] = (double)wm; red[wid][1] = (double)wsf; red[wid][2] = wlog; }
    __syncthreads();
    const double rm = red[w8][0], rs = red[w8][1];
    float M = (float)rm, q;
    q = tnn::dpp_move<0xB1, 0xf>(-INFINITY, M); M = q > M ? q : M;
    q = tnn::dpp_move<0x4E, 0xf>(-INFINITY, M); M = q > M ? q : M;
    q = tnn::dpp_move<0x141, 0xf>(-INFINITY, M); M = q > M ? q : M;
    double S = rm > -INFINITY ? rs * (double)expf((float)rm - M) : 0.0, L = red[w8][2];
    S += tnn::dpp_move<0xB1, 0xf>(0.0, S);
    S += tnn::dpp_move<0x4E, 0xf>(0.0, S);
    S += tnn::dpp_move<0x141, 0xf>(0.0, S);
    L += tnn::dpp_move<0xB1, 0xf>(0.0, L);
    L += tnn::dpp_move<0x4E, 0xf>(0.0, L);
    L += tnn::dpp_move<0x141, 0xf>(0.0, L);
    o.mx = mx; o.urow = urow; o.M = M; o.S = S; o.L = L;
}


// ---- the statistics at the TAIL of the forward launch that produced the partial logits ---------------------------------
// zpart [NP][m][C]: every tile column's share of the logits, written by the workgroups of THIS launch with system-scope
// (write-through) stores; the caller has established that all of them have landed (an agent-scope ticket, see
// dense_fwd_head_kernel) — they are read back here with cache-bypassing loads because this XCD's L2 may hold nothing
// (or stale lines) of what the other XCDs wrote.  512 threads: four per row, 128 rows at a time.
//   out_pair <- {M_r, S_r} of this shard, or, with `exchange`, the pair merged over all ranks of the peer-to-peer group
//   (tagged 8-byte stores, one link latency: ll_exchange2 on the loss kernel's slots and epoch counter — ONE workgroup per
//   launch runs this, so it may advance the counter itself).
struct HeadTail {
    unsigned int* ticket;            // agent-scope arrival counter of the launch (returns to 0 when the last one has arrived)
    const float* zpart;              // [NP][m][C]
    const float* bias;               // classifier bias [C]
    const float* y;                  // labels [m][C]
    float* out_pair;
    int m, exchange;
};

// {max, sum-exp} over rows [row_lo, row_hi) of the shard (all threads return the same pair)
template <int C, int NP>
__device__ __forceinline__ void head_tail_stats(const HeadTail& ta, const int row_lo, const int row_hi, float* zs, float* ys,
                                                double (*red)[4], float& M_out, float& S_out) {
    using namespace tnn::p2p;
    static_assert(NP == 8, "the partial-sum tree is written for 8 tiles");
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int mt = row_hi, stride = ta.m * C;        // mt: end of the row range; stride: the partial array's tile stride
    const bool vec = (ta.m % 2 == 0) && (row_hi % 2 == 0) && ((reinterpret_cast<uintptr_t>(ta.y) | reinterpret_cast<uintptr_t>(ta.zpart)) & 15) == 0;
    const int srow = t >> 2, sub = t & 3;
    // a range of more than 128 rows: blocks of 128 rows, one after the other, their {max, sum-exp} merged as the ranks' pairs are
    // (the forward launch gives every 128-row block to the workgroup that finishes it LAST, so ranges are one block there)
    float M = -INFINITY, S = 0.f;
    // vec path: the partials of TWO blocks are requested before the first wait (their round trips through the memory side —
    // ~2 us each at system scope — overlap), the third block's as soon as the first one's registers are free
    // (system-scope loads the COMPILER can see — buffer loads with the sc0 sc1 cache-policy bits — because they are carried
    // across loop iterations: a register copy the compiler inserts behind an inline-asm load reads the register before the
    // data has landed; measured, rows = 384 gave NaN statistics)
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    f32x4 cur[NP], nxt[NP], ycur = {0.f, 0.f, 0.f, 0.f}, ynxt = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ta.zpart), 0,
                                                                        (uint32_t)(NP * stride) * 4u, 0x00020000);
    auto issue = [&](f32x4 (&v)[NP], f32x4& yv, const int row0) {
        const int n = min(128, mt - row0) * C, base = row0 * C;
        if (t < (n >> 2)) {
#pragma unroll
            for (int tn = 0; tn < NP; ++tn)
                v[tn] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zr, (uint32_t)(base + 4 * t) * 4u,
                                                                                        (uint32_t)(tn * stride) * 4u, 17));
            yv = *reinterpret_cast<const f32x4*>(ta.y + base + 4 * t);
        }
    };
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};           // the bias of this thread's four staged elements: the same for every block
    if (vec) {
        issue(cur, ycur, row_lo);
        if (mt > row_lo + 128) issue(nxt, ynxt, row_lo + 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = ta.bias[(4 * t + i) % C];
    }
    for (int row0 = row_lo; row0 < mt; row0 += 128) {
        const int m = min(128, mt - row0), n = m * C, base = row0 * C;
        if (row0 > row_lo) __syncthreads();
        if (vec) {
            if (t < (n >> 2)) {
                f32x4 s = ((cur[0] + cur[1]) + (cur[2] + cur[3])) + ((cur[4] + cur[5]) + (cur[6] + cur[7]));
                s += bias4;
                *reinterpret_cast<f32x4*>(zs + 4 * t) = s;
                *reinterpret_cast<f32x4*>(ys + 4 * t) = ycur;
            }
            if (row0 + 128 < mt) {
#pragma unroll
                for (int tn = 0; tn < NP; ++tn) cur[tn] = nxt[tn];
                ycur = ynxt;
                if (row0 + 256 < mt) issue(nxt, ynxt, row0 + 256);
            }
        } else {
#pragma unroll 1
            for (int e = t; e < n; e += 512) {
                uint32_t u[NP];
#pragma unroll
                for (int tn = 0; tn < NP; ++tn) load_sys(u[tn], reinterpret_cast<const uint32_t*>(ta.zpart) + (size_t)tn * stride + base + e);
                loads_landed(u);
                float f[NP];
#pragma unroll
                for (int tn = 0; tn < NP; ++tn) f[tn] = __uint_as_float(u[tn]);
                zs[e] = (((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]))) + ta.bias[e % C];
                ys[e] = ta.y[base + e];
            }
        }
        __syncthreads();
        const bool slive = srow < m;
        const int sr = min(srow, m - 1);
        float zc[3], yc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            zc[i] = zs[sr * C + min(sub + 4 * i, C - 1)];
            yc[i] = ys[sr * C + min(sub + 4 * i, C - 1)];
        }
        HeadStats st;
        head_stats<C, false>(zc, yc, slive, sub, lane, wid, red, st);
        if (row0 == row_lo) { M = st.M; S = (float)st.S; }
        else {
            const float nm = fmaxf(M, st.M);
            S = S * expf(M - nm) + (float)st.S * expf(st.M - nm);
            M = nm;
        }
    }
    M_out = M;
    S_out = S;
}

// The shard's pair is complete in every thread of ONE workgroup: on the peer-to-peer transport exchange and merge it with the
// other ranks' (tagged 8-byte stores, one link latency), then leave it in out_pair for the head launch.
__device__ __forceinline__ void head_tail_finish(const HeadTail& ta, const tnn::p2p::LaunchCtx& ctx, float M, float S) {
    using namespace tnn::p2p;
    const int t = threadIdx.x;
    if (ta.exchange) {
        __shared__ float peer_stats[MAXW][2];
        const Peers& P = ctx.peers;
        const int W = P.world;
        const uint32_t ep = *ctx.ag_epoch;
        if (t < 2 * W) peer_stats[t >> 1][t & 1] = ll_exchange2(P, ep, (t & 1) ? S : M, ctx.dead, ctx.timeout_ticks);
        __syncthreads();
        float gm = -INFINITY, gs = 0.f;
        for (int q = 0; q < W; ++q) gm = fmaxf(gm, peer_stats[q][0]);
        for (int q = 0; q < W; ++q) gs += peer_stats[q][1] * expf(peer_stats[q][0] - gm);
        M = gm; S = gs;
        if (t == 0) *ctx.ag_epoch = ep + 1;
    }
    if (t == 0) { ta.out_pair[0] = M; ta.out_pair[1] = S; }
}

}  // namespace
