// Row-per-thread whole-batch softmax NLL (core/losses.py:24-32) as a device function, shared by the stand-alone loss
// kernel (tnn_fused.hip) and the persistent forward+backward kernel (tnn_mega.hip).
#pragma once
#include "tnn_internal.h"
#include "tnn_p2p.h"

namespace tnn {

// transcendental in the array's own precision (f32: hardware v_exp/v_log based expf/logf, <= 1 ulp; the sums
// they feed are still accumulated in f64); f64 arrays keep f64 throughout (exact-mode parity)
__device__ __forceinline__ double nll_exp(float x) { return (double)expf(x); }
__device__ __forceinline__ double nll_exp(double x) { return exp(x); }
__device__ __forceinline__ double nll_log(float x) { return (double)logf(x); }
__device__ __forceinline__ double nll_log(double x) { return log(x); }

// Classifier heads (c <= 16 classes, m <= 1024 rows): ONE THREAD PER ROW and a single block reduction.
// Each row is normalised against its OWN maximum first (m_i, e_ik = exp(z_ik - m_i), s_i = sum_k e_ik, u_i = sum_k e_ik y_ik),
// which needs no communication; the whole-batch quantities then follow from one combined reduction of
// {M = max m_i, S = sum s_i exp(m_i - M), L = sum (log u_i + m_i)}:
//     loss = log S + M - L / m,      dz_ik = e_ik exp(m_i - M) / S - (e_ik y_ik / u_i) / m
// The element-parallel kernel above needs three dependent block reductions (max, sum-exp, loss) with 16 waves; this one
// has one, over m / 64 waves — 4.5 -> ~3.3 us for the 128 x 10 MNIST head, whose cost is all dependent latency.
// SHARDED as above: the {M, S} pair is exchanged with the peers between the reduction and the dz pass.
// COH: z is read and dz written with agent-scope (sc1) accesses — for callers that run inside a persistent kernel
// whose other workgroups (on other XCDs, behind other L2s) produced z / will consume dz without a kernel boundary.
template <typename T, bool SHARDED, bool COH>
__device__ __forceinline__ void nll_rows_body(const T* __restrict__ z, const T* __restrict__ y, int m, int c,
                                              T* __restrict__ stats_out, T* __restrict__ loss_out,
                                              T* __restrict__ dz, double inv_m_global,
                                              const tnn::p2p::LaunchCtx& ctx, double* __restrict__ tick, double b1,
                                              double b2) {
    constexpr int CMAX = 16;
    __shared__ double wave_m[16], wave_s[16], wave_l[16];
    __shared__ double scal[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = (blockDim.x + 63) >> 6;
    if (tick != nullptr && tid == 0) {
        tick[0] *= b1;
        tick[1] *= b2;
    }
    T e[CMAX], ey[CMAX];
    double mi = -INFINITY, si = 0.0, li = 0.0, ui = 1.0;
    const bool live = tid < m;
    // f32 batches above 128 rows: z and y are flat [m][c] arrays, so the workgroup first copies them into LDS with coalesced
    // 16-B loads (5 per thread at 1024 x 10) and every thread then picks its row there.  Row-wise from global memory each
    // load instruction of a wave touches ~20 cache lines, and this kernel is ONE workgroup on ONE CU: at 1024 rows the
    // 20 x 16 wave-loads were most of its 11.4 us.
    constexpr bool kStage = sizeof(T) == 4 && !COH;
    constexpr int kStageElems = kStage ? 1024 * 10 : 4;
    __shared__ __attribute__((aligned(16))) float zst[kStageElems], yst[kStageElems];
    bool staged = false;
    if constexpr (kStage) {
        const int n = m * c;
        staged = m > 128 && n <= kStageElems && (n & 3) == 0 &&
                 ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0;      // block-uniform
        if (staged) {
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            for (int i = tid; i < (n >> 2); i += blockDim.x) {
                *reinterpret_cast<f32x4_t*>(zst + 4 * i) = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(z) + 4 * i);
                *reinterpret_cast<f32x4_t*>(yst + 4 * i) = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(y) + 4 * i);
            }
            __syncthreads();
        }
    }
    if (live) {
        T zr[CMAX], yr[CMAX];
        if (staged) {
#pragma unroll
            for (int k = 0; k < CMAX; ++k) {
                zr[k] = k < c ? (T)zst[tid * c + k] : (T)-INFINITY;
                yr[k] = k < c ? (T)yst[tid * c + k] : (T)0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < CMAX; ++k) {
                zr[k] = k < c ? (COH ? __hip_atomic_load(z + (int64_t)tid * c + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : z[(int64_t)tid * c + k])
                              : (T)-INFINITY;
                yr[k] = k < c ? y[(int64_t)tid * c + k] : (T)0;
            }
        }
        T mx = zr[0];
#pragma unroll
        for (int k = 1; k < CMAX; ++k) mx = zr[k] > mx ? zr[k] : mx;
        double s = 0.0, u = 0.0;
#pragma unroll
        for (int k = 0; k < CMAX; ++k) {
            const double ek = k < c ? nll_exp((T)(zr[k] - mx)) : 0.0;
            e[k] = (T)ek;
            ey[k] = (T)((double)(T)ek * (double)yr[k]);
            s += ek;
            u += (double)ey[k];
        }
        mi = (double)mx; si = s; ui = u;
        li = nll_log((T)u) + (double)mx;
    }
    // one combined reduction: waves first (DPP steps — a 64-bit shuffle tree costs ~700 cycles, three of them per wave and
    // up to 16 waves sharing one CU here), then <= 16 wave triples through LDS.  All 64 lanes of every wave are active (the
    // kernel is launched with whole waves; rows beyond m carry the identities).
    double wm = wave_max_dpp(mi);
    double ws = wave_sum_dpp(live ? si * nll_exp((T)(mi - wm)) : 0.0);
    double wl = wave_sum_dpp(li);
    if (lane == 0) { wave_m[wid] = wm; wave_s[wid] = ws; wave_l[wid] = wl; }
    __syncthreads();
    // every lane takes wave entry (lane & 15) and a 16-lane DPP butterfly leaves M, S, L in ALL lanes: one exp per lane instead
    // of a serial loop over up to 16 entries in every thread (1.4 us of the 1024-row kernel)
    const int w16 = lane & 15;
    const bool have = w16 < nw;
    const double rm = have ? wave_m[w16] : -INFINITY, rs = have ? wave_s[w16] : 0.0, rl = have ? wave_l[w16] : 0.0;
    double M = rm, q;
    q = dpp_move<0xB1, 0xf>((double)-INFINITY, M); M = fmax(M, q);
    q = dpp_move<0x4E, 0xf>((double)-INFINITY, M); M = fmax(M, q);
    q = dpp_move<0x141, 0xf>((double)-INFINITY, M); M = fmax(M, q);
    q = dpp_move<0x140, 0xf>((double)-INFINITY, M); M = fmax(M, q);
    double S = rm > -INFINITY ? rs * nll_exp((T)(rm - M)) : 0.0, L = rl;
    S += dpp_move<0xB1, 0xf>(0.0, S); L += dpp_move<0xB1, 0xf>(0.0, L);
    S += dpp_move<0x4E, 0xf>(0.0, S); L += dpp_move<0x4E, 0xf>(0.0, L);
    S += dpp_move<0x141, 0xf>(0.0, S); L += dpp_move<0x141, 0xf>(0.0, L);
    S += dpp_move<0x140, 0xf>(0.0, S); L += dpp_move<0x140, 0xf>(0.0, L);
    double inv_m = 1.0 / (double)m;
    double loss;
    if constexpr (SHARDED) {
        using namespace tnn::p2p;
        __shared__ float peer_stats[MAXW][2];
        const Peers& P = ctx.peers;
        const int W = P.world;
        const uint32_t ep = *ctx.ag_epoch;
        if (tid < 2 * W)           // {M_r, S_r} to every rank and theirs back: one 8-byte tagged store per word and peer
            peer_stats[tid >> 1][tid & 1] =
                ll_exchange2(P, ep, (tid & 1) ? (float)S : (float)M, ctx.dead, ctx.timeout_ticks);
        __syncthreads();
        double gm = -INFINITY, gs = 0.0;
        for (int q = 0; q < W; ++q) gm = fmax(gm, (double)peer_stats[q][0]);
        for (int q = 0; q < W; ++q) gs += (double)peer_stats[q][1] * nll_exp((T)((double)peer_stats[q][0] - gm));
        if (tid == 0) *ctx.ag_epoch = ep + 1;
        M = gm; S = gs;
        inv_m = inv_m_global;
        loss = ((nll_log((T)S) + M) * (double)m - L) * inv_m;   // this rank's share
    } else {
        loss = nll_log((T)S) + M - L * inv_m;
    }
    if constexpr (kStage) {
        if (staged && dz) {                   // block-uniform: dz leaves the way z came in — through LDS, coalesced 16-B stores
            if (live) {
                const float sf = (float)(nll_exp((T)(mi - M)) / S), uf = (float)(inv_m / ui);
#pragma unroll
                for (int k = 0; k < CMAX; ++k)
                    if (k < c) zst[tid * c + k] = (float)e[k] * sf - (float)ey[k] * uf;
            }
            __syncthreads();
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            for (int i = tid; i < ((m * c) >> 2); i += blockDim.x)
                *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(dz) + 4 * i) = *reinterpret_cast<const f32x4_t*>(zst + 4 * i);
        }
    }
    if (live && dz && !staged) {
        const double scale = nll_exp((T)(mi - M)) / S, inv_u = inv_m / ui;
        if (sizeof(T) == 4) {
            const float sf = (float)scale, uf = (float)inv_u;
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < c) {
                    const T dv = (T)((float)e[k] * sf - (float)ey[k] * uf);
                    if (COH) __hip_atomic_store(dz + (int64_t)tid * c + k, dv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else dz[(int64_t)tid * c + k] = dv;
                }
        } else {
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < c) {
                    const T dv = (T)((double)e[k] * scale - (double)ey[k] * inv_u);
                    if (COH) __hip_atomic_store(dz + (int64_t)tid * c + k, dv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else dz[(int64_t)tid * c + k] = dv;
                }
        }
    }
    if (tid == 0) {
        if (loss_out) loss_out[0] = (T)loss;
        if (stats_out) { stats_out[0] = (T)M; stats_out[1] = (T)S; }
    }
}

}  // namespace tnn
