// Fused hot-path kernels: bias(+ReLU) (K8), whole-batch softmax-NLL forward+backward (K9),
// SGD / Adam on the flat parameter arena (K10).
#include <math.h>

#include "tnn_internal.h"
#include "tnn_p2p.h"
#include "tnn_nll_rows.h"

namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------------------ bias + activation
// y[r, c] = act(x[r, c] + b[c]).  HBM-bound, 8 B/element; float4 along the row when N % 4 == 0.
template <typename T, int ACT, int VEC>
__global__ __launch_bounds__(kThreads) void bias_act_kernel(const T* __restrict__ x,
                                                            const T* __restrict__ bias,
                                                            T* __restrict__ y, int64_t M,
                                                            int64_t N) {
    int64_t nv = N / VEC;
    for (int64_t r = blockIdx.y; r < M; r += gridDim.y) {
        const T* xr = x + r * N;
        T* yr = y + r * N;
        for (int64_t cv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cv < nv;
             cv += (int64_t)gridDim.x * blockDim.x) {
            T v[VEC], b[VEC];
            if constexpr (VEC == 4) {
                float4 t = *reinterpret_cast<const float4*>(xr + cv * 4);
                float4 u = *reinterpret_cast<const float4*>(bias + cv * 4);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                b[0] = u.x; b[1] = u.y; b[2] = u.z; b[3] = u.w;
            } else {
                v[0] = xr[cv];
                b[0] = bias[cv];
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                v[k] = v[k] + b[k];
                if (ACT == TNN_ACT_RELU) v[k] = v[k] < T(0) ? T(0) : v[k];   // clip(x, 0.0)
            }
            if constexpr (VEC == 4) {
                *reinterpret_cast<float4*>(yr + cv * 4) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                yr[cv] = v[0];
            }
        }
    }
}

template <typename T, int ACT>
int bias_act_typed(const void* x, const void* bias, void* y, int64_t M, int64_t N) {
    bool vec = sizeof(T) == 4 && (N % 4 == 0) &&
               ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                 reinterpret_cast<uintptr_t>(bias)) & 15) == 0;
    int64_t per_row = vec ? N / 4 : N;
    unsigned gx = (unsigned)((per_row + kThreads - 1) / kThreads);
    int64_t cap = (int64_t)tnn::num_cus() * 8;
    if (gx > cap) gx = (unsigned)cap;
    int64_t gy = cap / gx;
    if (gy > M) gy = M;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    if constexpr (sizeof(T) == 4) {
        if (vec) {
            hipLaunchKernelGGL((bias_act_kernel<T, ACT, 4>), dim3(gx, (unsigned)gy), kThreads, 0,
                               tnn::stream(), (const T*)x, (const T*)bias, (T*)y, M, N);
            TNN_LAUNCH_OK();
            return 0;
        }
    }
    hipLaunchKernelGGL((bias_act_kernel<T, ACT, 1>), dim3(gx, (unsigned)gy), kThreads, 0,
                       tnn::stream(), (const T*)x, (const T*)bias, (T*)y, M, N);
    TNN_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------------------------------ softmax NLL (F5)
// core/losses.py:24-32 normalises over the WHOLE [m, c] batch:  M = max z,  S = sum exp(z - M).
// Stage 1: per-block (M_b, S_b); stage 2 merges with S = sum S_b * exp(M_b - M).  Both in f64.
// final != NULL (single-block launches): the block's pair IS the result and goes straight out as T.
template <typename T>
__global__ __launch_bounds__(kThreads) void nll_stats_kernel(const T* __restrict__ z, int64_t n,
                                                             double* __restrict__ partial, T* __restrict__ final) {
    __shared__ double lds_m[kThreads / 64], lds_s[kThreads / 64];
    double mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double v = (double)z[i];
        mx = v > mx ? v : mx;
    }
    mx = tnn::wave_max(mx);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds_m[w] = mx;
    __syncthreads();
    double bm = lds_m[0];
#pragma unroll
    for (int i = 1; i < kThreads / 64; ++i) bm = lds_m[i] > bm ? lds_m[i] : bm;
    double s = 0.0;
    if (bm > -INFINITY) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            s += exp((double)z[i] - bm);
    }
    s = tnn::wave_sum(s);
    if (lane == 0) lds_s[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kThreads / 64; ++i) t += lds_s[i];
        if (final) {
            final[0] = (T)bm;
            final[1] = (T)t;
        } else {
            partial[2 * blockIdx.x] = bm;
            partial[2 * blockIdx.x + 1] = t;
        }
    }
}

// stats_all [n,2] (TI) -> stats [2] (TO); one wave
template <typename TI, typename TO>
__global__ __launch_bounds__(64) void lse_merge_kernel(const TI* __restrict__ all, int n,
                                                       TO* __restrict__ out) {
    int lane = threadIdx.x;
    double mx = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        double v = (double)all[2 * i];
        mx = v > mx ? v : mx;
    }
    mx = tnn::wave_max(mx);
    double s = 0.0;
    for (int i = lane; i < n; i += 64) {
        double mi = (double)all[2 * i], si = (double)all[2 * i + 1];
        if (mi > -INFINITY) s += si * exp(mi - mx);
    }
    s = tnn::wave_sum(s);
    if (lane == 0) {
        out[0] = (TO)mx;
        out[1] = (TO)s;
    }
}

// one thread per row:  e = exp(z - M), q = sum_k e*y, nll = log S - log q,
// dz = e/S - (e*y/q)/m_global;  block partial of nll/m_global -> partial[blockIdx.x]
template <typename T>
__global__ __launch_bounds__(kThreads) void nll_fwd_bwd_kernel(const T* __restrict__ z,
                                                               const T* __restrict__ y, int64_t m,
                                                               int64_t c, double inv_m_global,
                                                               const T* __restrict__ stats,
                                                               double* __restrict__ partial,
                                                               T* __restrict__ dz, T* __restrict__ final_loss) {
    __shared__ double lds[kThreads / 64];
    const double M = (double)stats[0], S = (double)stats[1];
    const double log_s = log(S), inv_s = 1.0 / S;
    double local = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m;
         r += (int64_t)gridDim.x * blockDim.x) {
        const T* zr = z + r * c;
        const T* yr = y + r * c;
        double q = 0.0;
        for (int64_t k = 0; k < c; ++k) q += exp((double)zr[k] - M) * (double)yr[k];
        local += (log_s - log(q)) * inv_m_global;
        if (dz) {
            T* dr = dz + r * c;
            double inv_q = inv_m_global / q;
            for (int64_t k = 0; k < c; ++k) {
                double e = exp((double)zr[k] - M);
                dr[k] = (T)(e * inv_s - e * (double)yr[k] * inv_q);
            }
        }
    }
    local = tnn::wave_sum(local);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds[w] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kThreads / 64; ++i) t += lds[i];
        if (final_loss) final_loss[0] = (T)t;        // single-block launch: the partial is the loss
        else if (partial) partial[blockIdx.x] = t;
    }
}


// Single-launch whole-batch softmax NLL for small batches (the MNIST head is 128 x 10): one 1024-thread
// block, element-parallel.  exp(z - M) is evaluated ONCE per element and kept in LDS; the per-row q and the
// three block reductions (max, sum-exp, loss) go through wave shuffles + one LDS hop each.
constexpr int kNllMaxRows = 1024;
// transcendental in the array's own precision (f32: hardware v_exp/v_log based expf/logf, <= 1 ulp; the sums
// they feed are still accumulated in f64); f64 arrays keep f64 throughout (exact-mode parity)
using tnn::nll_exp;   // tnn_nll_rows.h
using tnn::nll_log;
template <typename T> struct NllCap { static constexpr int elems = sizeof(T) == 4 ? 4096 : 2048; };

// Block reduction tuned for latency (measured on MI355X: a 64-bit __shfl_xor is two ds_bpermute round trips
// per step and a two-level tree needs three barriers — ~1700 cycles per reduction, and the kernel needs
// three).  Here the cross-lane tree runs on R (float for f32 arrays: one ds_bpermute per step), every wave
// leaves one partial in its own LDS slot, ONE barrier follows, and every thread then sums the <= 16 partials
// itself in f64 (broadcast LDS reads) — no second tree, no broadcast barrier.  Each reduction of a kernel
// uses its own slot array, so no barrier is needed to recycle them.
template <bool IS_MAX, typename R>
__device__ __forceinline__ double block_reduce_fast(R v, R* slots) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        R other = __shfl_xor(v, o, 64);
        v = IS_MAX ? (other > v ? other : v) : v + other;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) slots[w] = v;
    __syncthreads();
    // all 16 slots are fetched with independent loads issued back to back (a dependent 16-iteration loop of
    // LDS reads alone measured ~1100 cycles); slots beyond nw hold the reduction's identity
    R part[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) part[i] = slots[i];
    double r = (double)part[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const double x = (double)part[i];
        if (i < nw) r = IS_MAX ? (x > r ? x : r) : r + x;
    }
    return r;
}

// z and y are fetched ONCE, up front and coalesced, into LDS; every later phase touches only LDS, so the
// kernel has a single global-memory round trip before its final store.
//
// SHARDED (data parallel, f32): m is this rank's row count and the softmax still spans the GLOBAL batch
// (core/losses.py:26-27).  After the local {M_r, S_r} the workgroup pushes the pair into every peer's all-gather
// slot over xGMI (csrc/tnn_p2p.h), meets the peers' workgroups at a flag barrier and merges
// M = max_r M_r, S = sum_r S_r exp(M_r - M).  Nothing else is recomputed: with f = exp(M_r - M) the global
// probabilities are e_local * f / S, the term (e*y)/q is invariant under the common factor f, and
// log q_global = log q_local + (M_r - M) — so only two scalars change.  loss_out receives this rank's SHARE
// (rows_local/m_global * log-normaliser - sum(log q)/m_global): the all-reduce of the gradient arena sums it.
template <typename T, bool SHARDED>
__global__ __launch_bounds__(1024) void nll_fused_kernel(const T* __restrict__ z, const T* __restrict__ y,
                                                         int m, int c, T* __restrict__ stats_out,
                                                         T* __restrict__ loss_out, T* __restrict__ dz,
                                                         double inv_m_global, tnn::p2p::LaunchCtx ctx,
                                                         double* __restrict__ tick, double b1, double b2) {
    constexpr int kMax = NllCap<T>::elems;
    __shared__ T e_lds[kMax];              // z -> exp(z - M)
    __shared__ T y_lds[kMax];              // y -> e * y
    __shared__ double q_lds[kNllMaxRows];
    __shared__ T red_max[16], red_sum[16], red_loss[16];
    const int tid = threadIdx.x, n = m * c;
    // the loss kernel is the one single-workgroup launch of a training step: its thread 0 can also advance Adam's
    // {b1^t, b2^t} (tick != NULL), which otherwise costs a launch of its own (adam_advance_kernel, ~1.6 us)
    if (tick != nullptr && tid == 0) {
        tick[0] *= b1;
        tick[1] *= b2;
    }
    T mx = -INFINITY;
    for (int i = tid; i < n; i += blockDim.x) {
        const T zi = z[i];
        y_lds[i] = y[i];
        e_lds[i] = zi;
        mx = zi > mx ? zi : mx;
    }
    const double M = block_reduce_fast<true, T>(mx, red_max);   // exact: a max of T values
    const T Mt = (T)M;
    double s = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
        const double e = nll_exp((T)(e_lds[i] - Mt));
        e_lds[i] = (T)e;
        y_lds[i] = (T)((double)(T)e * (double)y_lds[i]);
        s += e;
    }
    const double S = block_reduce_fast<false, T>((T)s, red_sum);    // its barrier also publishes e/y_lds
    __shared__ double scal[4];
    if constexpr (SHARDED) {
        using namespace tnn::p2p;
        __shared__ float peer_stats[MAXW][2];
        const Peers& P = ctx.peers;
        const int W = P.world;
        const uint32_t ep = *ctx.ag_epoch;
        const size_t slots = offsetof(Header, ag_slot) + (size_t)(ep & 1) * MAXW * AG_BYTES;
        if (tid < 2 * W) {                                            // {M_r, S_r} -> slot [rank] of every peer
            const float mine = (tid & 1) ? (float)S : (float)M;
            store_sys(reinterpret_cast<uint32_t*>(P.base[tid >> 1] + slots + (size_t)P.rank * AG_BYTES) + (tid & 1),
                      __float_as_uint(mine));
        }
        exchange_flags(P, offsetof(Header, ag_flag), ep + 1, ctx.dead, ctx.timeout_ticks);
        if (tid < 2 * W) {
            uint32_t w[1] = {0u};
            load_sys(w[0], reinterpret_cast<const uint32_t*>(P.base[P.rank] + slots + (size_t)(tid >> 1) * AG_BYTES) + (tid & 1));
            loads_landed(w);
            peer_stats[tid >> 1][tid & 1] = __uint_as_float(w[0]);
        }
        __syncthreads();
        if (tid == 0) {
            double gm = -INFINITY, gs = 0.0;
            for (int q = 0; q < W; ++q) gm = fmax(gm, (double)peer_stats[q][0]);
            for (int q = 0; q < W; ++q) gs += (double)peer_stats[q][1] * exp((double)peer_stats[q][0] - gm);
            const double shift = M - gm;                               // <= 0
            scal[0] = log(gs) - shift;                                 // per-row log-normaliser for LOCAL q
            scal[1] = exp(shift) / gs;                                 // e_local -> global probability
            scal[2] = gm;
            scal[3] = gs;
            *ctx.ag_epoch = ep + 1;
        }
    } else {
        if (tid == 0) { scal[0] = log(S); scal[1] = 1.0 / S; scal[2] = M; scal[3] = S; }   // once, not once per wave
    }
    const double inv_m = SHARDED ? inv_m_global : 1.0 / (double)m;
    double local = 0.0;
    for (int r = tid; r < m; r += blockDim.x) {
        double q = 0.0;
        if (c <= 16) {                       // classifier heads: the row's values in independent LDS reads
            T row[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) row[k] = k < c ? y_lds[r * c + k] : T(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) q += (double)row[k];
        } else {
            for (int k = 0; k < c; ++k) q += (double)y_lds[r * c + k];
        }
        q_lds[r] = sizeof(T) == 4 ? (double)((float)inv_m / (float)q) : inv_m / q;
        local -= nll_log((T)q);
    }
    const double sum_log_q = block_reduce_fast<false, T>((T)local, red_loss);   // barrier publishes q_lds, scal
    // log S - mean(log q); a shard contributes its rows' share of the log-normaliser
    const double loss = (SHARDED ? scal[0] * (double)m * inv_m : scal[0]) + sum_log_q * inv_m;
    const double inv_s = scal[1];
    if (dz) {
        if (sizeof(T) == 4) {
            const float inv_sf = (float)inv_s;
            for (int i = tid; i < n; i += blockDim.x)
                dz[i] = (T)((float)e_lds[i] * inv_sf - (float)y_lds[i] * (float)q_lds[i / c]);
        } else {
            for (int i = tid; i < n; i += blockDim.x)
                dz[i] = (T)((double)e_lds[i] * inv_s - (double)y_lds[i] * q_lds[i / c]);
        }
    }
    if (tid == 0) {
        if (loss_out) loss_out[0] = (T)loss;
        if (stats_out) { stats_out[0] = (T)scal[2]; stats_out[1] = (T)scal[3]; }
    }
}

template <typename T, bool SHARDED>
__global__ __launch_bounds__(1024) void nll_rows_kernel(const T* __restrict__ z, const T* __restrict__ y, int m, int c,
                                                        T* __restrict__ stats_out, T* __restrict__ loss_out,
                                                        T* __restrict__ dz, double inv_m_global,
                                                        tnn::p2p::LaunchCtx ctx, double* __restrict__ tick, double b1,
                                                        double b2) {
    tnn::nll_rows_body<T, SHARDED, false>(z, y, m, c, stats_out, loss_out, dz, inv_m_global, ctx, tick, b1, b2);
}

template <typename TO>
__global__ __launch_bounds__(64) void sum_partials_kernel(const double* __restrict__ partial, int n,
                                                          TO* __restrict__ out, TO* __restrict__ out2 = nullptr) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = tnn::wave_sum(s);
    if (threadIdx.x == 0) {
        out[0] = (TO)s;
        if (out2) out2[0] = (TO)s;               // e.g. the step's slot of a loss history: no copy launch
    }
}

// sum((pred-y)^2)/m and its gradient; block partials in f64, combined by sum_partials_kernel
template <typename T>
__global__ __launch_bounds__(kThreads) void mse_fwd_bwd_kernel(const T* __restrict__ pred,
                                                               const T* __restrict__ y, int64_t n,
                                                               double inv_m,
                                                               double* __restrict__ partial,
                                                               T* __restrict__ dpred, double* __restrict__ tick = nullptr,
                                                               double b1 = 1.0, double b2 = 1.0, const int* guard = nullptr) {
    __shared__ double lds[kThreads / 64];
    // Adam's {b1^t, b2^t} advanced by ONE thread of the step's loss kernel (nothing in this launch reads them; the optimizer
    // launches behind it do): saves the one-thread launch of tnn_adam_tick
    if (tick != nullptr && blockIdx.x == 0 && threadIdx.x == 0 &&
        (guard == nullptr || __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)) {
        tick[0] *= b1;
        tick[1] *= b2;
    }
    double local = 0.0;
    const T two_inv_m = (T)(2.0 * inv_m);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        T e = pred[i] - y[i];
        local += (double)e * (double)e;
        if (dpred) dpred[i] = two_inv_m * e;
    }
    local = tnn::wave_sum(local);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds[w] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kThreads / 64; ++i) t += lds[i];
        partial[blockIdx.x] = t * inv_m;
    }
}

// ------------------------------------------------------------------------------ SGD / Adam
template <typename T>
__global__ __launch_bounds__(kThreads) void sgd_kernel(T* __restrict__ p, const T* __restrict__ g,
                                                       int64_t n, T lr, const int* guard) {
    TNN_GUARD_RETURN(guard);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        p[i] = p[i] + (-lr * g[i]);
}

// Momentum / RMSProp / Adagrad / Adadelta (core/optimizer.py:82-164), one pass each over the flat arena: read g and
// the optimizer's one or two state vectors, write the state and the step (and p += step when p is given).
// The expressions keep the reference's operation order (e.g. lr * g / sqrt(ms + eps), not g * (lr / sqrt(..))).
template <typename T, int KIND>
__global__ __launch_bounds__(kThreads) void optim_kernel(T* __restrict__ p, const T* __restrict__ g,
                                                         T* __restrict__ s1, T* __restrict__ s2,
                                                         T* __restrict__ step_out, int64_t n, T lr, T a, T b, T eps,
                                                         const int* guard) {
    TNN_GUARD_RETURN(guard);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const T gi = g[i];
        const T pi = p ? p[i] : T(0);                        // requested with the other streams, not after the maths
        T step;
        if constexpr (KIND == TNN_OPT_MOMENTUM) {            // acc = momentum * acc + g; step = -lr * acc
            const T acc = a * s1[i] + gi;
            s1[i] = acc;
            step = -lr * acc;
        } else if constexpr (KIND == TNN_OPT_RMSPROP) {      // ms += (1-decay)(g^2 - ms); mom = momentum*mom + lr*g/sqrt(ms+eps)
            T ms = s1[i];
            ms = ms + (T(1) - a) * (gi * gi - ms);
            s1[i] = ms;
            const T mom = b * s2[i] + lr * gi / sqrt(ms + eps);
            s2[i] = mom;
            step = -mom;
        } else if constexpr (KIND == TNN_OPT_ADAGRAD) {      // G += g^2; step = -(lr / sqrt(G + eps)) * g
            const T G = s1[i] + gi * gi;
            s1[i] = G;
            step = -(lr / sqrt(G + eps)) * gi;
        } else {                                             // Adadelta
            T Eg = s1[i];
            Eg = Eg + (T(1) - a) * (gi * gi - Eg);
            s1[i] = Eg;
            T d = s2[i];
            const T delta = gi * (sqrt(d + eps) / sqrt(Eg + eps));
            step = -lr * delta;
            s2[i] = d + (T(1) - a) * (delta * delta - d);
        }
        if (step_out) step_out[i] = step;
        if (p) p[i] = pi + step;
    }
}

// Adam, core/optimizer.py:67-79, one pass: read p,g,m,v / write p,m,v = 28 B per fp32 parameter.
// state (device, f64): [0] = b1^t, [1] = b2^t of the CURRENT step, advanced on the device by
// adam_advance_kernel (one thread) right before this kernel, so that a captured hipGraph replays the right
// bias correction without any host-side step counter.  (A first version advanced the state inside this
// kernel with an atomic ticket per block: 230 same-address atomics cost ~3 us, more than the whole update.)
// The same single thread can carry one scalar along (esz bytes, 4 or 8): the data-parallel trainer files the
// all-reduced loss into its loss history this way instead of paying a separate copy launch per step.
__global__ void adam_advance_kernel(double* __restrict__ state, double b1, double b2, const void* __restrict__ src,
                                    void* __restrict__ dst, int esz, const int* guard) {
    TNN_GUARD_RETURN(guard);
    state[0] *= b1;
    state[1] *= b2;
    if (dst) {
        if (esz == 8) *(double*)dst = *(const double*)src;
        else *(float*)dst = *(const float*)src;
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void adam_kernel(T* __restrict__ p, const T* __restrict__ g,
                                                        T* __restrict__ m, T* __restrict__ v,
                                                        int64_t n, T lr, T b1, T b2, T eps,
                                                        const double* __restrict__ state,
                                                        T* __restrict__ step_out, const int* guard,
                                                        const T* __restrict__ scalar_src = nullptr,
                                                        T* __restrict__ scalar_dst = nullptr,
                                                        double* __restrict__ self_advance = nullptr,
                                                        double ab1 = 1.0, double ab2 = 1.0) {
    TNN_GUARD_RETURN(guard);
    // one scalar rides along (the data-parallel trainer files the all-reduced loss into its loss history): no prologue
    // launch just for a 4-byte copy
    if (scalar_dst != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *scalar_dst = *scalar_src;
    // self_advance (== state, small grids only): this launch ALSO advances the beta powers — every workgroup works with
    // state * {b1, b2} computed from the values it read, and the workgroup that finishes LAST (agent-scope arrival counter
    // in state[2]) stores them: nobody can read a value that was advanced twice, and the one-thread prologue launch is gone.
    // (Round 1 drew the ticket at the START of every block, 230 of them: ~3 us; <= 128 arrivals spread over the END of the launch.)
    const double p1 = state[0] * ab1, p2 = state[1] * ab2;
    const T inv_c1 = (T)(1.0 / (1.0 - p1)), inv_c2 = (T)(1.0 / (1.0 - p2));
    const T one_m_b1 = T(1) - b1, one_m_b2 = T(1) - b2;
    int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nth = (int64_t)gridDim.x * blockDim.x;
    auto upd = [&](T gi, T& mi, T& vi) -> T {
        mi = mi + one_m_b1 * (gi - mi);
        vi = vi + one_m_b2 * (gi * gi - vi);
        T mh = mi * inv_c1, vh = vi * inv_c2;
        return -lr * mh / (sqrt(vh) + eps);
    };
    if constexpr (VEC == 4) {
        int64_t nv = n / 4;
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (int64_t i = tid; i < nv; i += nth) {
            // all four streams requested before the first use (p used to be fetched after the arithmetic); g, m, v
            // are touched once per step: non-temporal, so they do not evict the parameters the next forward re-reads
            const f4 g4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i);
            f4 m4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i);
            f4 v4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i);
            f4 p4 = {0.f, 0.f, 0.f, 0.f};
            if (!step_out) p4 = reinterpret_cast<const f4*>(p)[i];
            f4 s4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float mk = m4[k], vk = v4[k];
                s4[k] = upd(g4[k], mk, vk);
                m4[k] = mk; v4[k] = vk;
            }
            __builtin_nontemporal_store(m4, reinterpret_cast<f4*>(m) + i);
            __builtin_nontemporal_store(v4, reinterpret_cast<f4*>(v) + i);
            if (step_out) reinterpret_cast<f4*>(step_out)[i] = s4;
            else reinterpret_cast<f4*>(p)[i] = p4 + s4;
        }
        for (int64_t i = nv * 4 + tid; i < n; i += nth) {
            T mi = m[i], vi = v[i];
            T s = upd(g[i], mi, vi);
            m[i] = mi; v[i] = vi;
            if (step_out) step_out[i] = s; else p[i] = p[i] + s;
        }
    } else {
        for (int64_t i = tid; i < n; i += nth) {
            T mi = m[i], vi = v[i];
            T s = upd(g[i], mi, vi);
            m[i] = mi; v[i] = vi;
            if (step_out) step_out[i] = s; else p[i] = p[i] + s;
        }
    }
    if (self_advance != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int* ticket = reinterpret_cast<unsigned int*>(self_advance + 2);
            const unsigned int prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) {
                self_advance[0] = p1;
                self_advance[1] = p2;
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace

extern "C" {

int tnn_bias_act(const void* x, const void* bias, int act, void* y, int64_t M, int64_t N, int dtype) {
    TNN_NEED_INIT();
    if (M * N <= 0) return 0;
    TNN_REQUIRE(act == TNN_ACT_NONE || act == TNN_ACT_RELU, "tnn_bias_act: unknown activation %d", act);
    switch (dtype) {
        case TNN_F32:
            return act == TNN_ACT_RELU ? bias_act_typed<float, TNN_ACT_RELU>(x, bias, y, M, N)
                                       : bias_act_typed<float, TNN_ACT_NONE>(x, bias, y, M, N);
        case TNN_F64:
            return act == TNN_ACT_RELU ? bias_act_typed<double, TNN_ACT_RELU>(x, bias, y, M, N)
                                       : bias_act_typed<double, TNN_ACT_NONE>(x, bias, y, M, N);
    }
    tnn::set_error("tnn_bias_act: dtype %d is not a float type", dtype);
    return 2;
}

int tnn_softmax_nll_stats(const void* z, int64_t m, int64_t c, void* stats, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(dtype == TNN_F32 || dtype == TNN_F64, "tnn_softmax_nll_stats: dtype %d", dtype);
    int64_t n = m * c;
    TNN_REQUIRE(n > 0, "tnn_softmax_nll_stats: empty logits");
    int64_t nb = (n + (int64_t)kThreads * 8 - 1) / ((int64_t)kThreads * 8);
    if (nb > 1024) nb = 1024;
    hipStream_t s = tnn::stream();
    if (nb == 1) {                                   // classifier-size logits: one block, one launch
        if (dtype == TNN_F32)
            hipLaunchKernelGGL((nll_stats_kernel<float>), 1, kThreads, 0, s, (const float*)z, n, (double*)nullptr, (float*)stats);
        else
            hipLaunchKernelGGL((nll_stats_kernel<double>), 1, kThreads, 0, s, (const double*)z, n, (double*)nullptr, (double*)stats);
        TNN_LAUNCH_OK();
        return 0;
    }
    void* ws = nullptr;
    if (tnn_malloc((size_t)nb * 2 * sizeof(double), &ws)) return 1;
    if (dtype == TNN_F32) {
        hipLaunchKernelGGL((nll_stats_kernel<float>), (unsigned)nb, kThreads, 0, s, (const float*)z, n, (double*)ws, (float*)nullptr);
        hipLaunchKernelGGL((lse_merge_kernel<double, float>), 1, 64, 0, s, (const double*)ws, (int)nb, (float*)stats);
    } else {
        hipLaunchKernelGGL((nll_stats_kernel<double>), (unsigned)nb, kThreads, 0, s, (const double*)z, n, (double*)ws, (double*)nullptr);
        hipLaunchKernelGGL((lse_merge_kernel<double, double>), 1, 64, 0, s, (const double*)ws, (int)nb, (double*)stats);
    }
    tnn_free(ws);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_lse_merge(const void* stats_all, int n_shards, void* stats, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(n_shards > 0, "tnn_lse_merge: n_shards %d", n_shards);
    hipStream_t s = tnn::stream();
    switch (dtype) {
        case TNN_F32: hipLaunchKernelGGL((lse_merge_kernel<float, float>), 1, 64, 0, s, (const float*)stats_all, n_shards, (float*)stats); break;
        case TNN_F64: hipLaunchKernelGGL((lse_merge_kernel<double, double>), 1, 64, 0, s, (const double*)stats_all, n_shards, (double*)stats); break;
        default: tnn::set_error("tnn_lse_merge: dtype %d is not a float type", dtype); return 2;
    }
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_softmax_nll_fwd_bwd(const void* z, const void* y, int64_t m, int64_t c, int64_t m_global,
                            const void* stats, void* loss_out, void* dz, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(dtype == TNN_F32 || dtype == TNN_F64, "tnn_softmax_nll_fwd_bwd: dtype %d", dtype);
    TNN_REQUIRE(m > 0 && c > 0 && m_global > 0, "tnn_softmax_nll_fwd_bwd: empty batch");
    int64_t nb = (m + kThreads - 1) / kThreads;
    if (nb > 1024) nb = 1024;
    hipStream_t s = tnn::stream();
    double inv_m = 1.0 / (double)m_global;
    if (nb == 1) {                                   // up to 256 rows: one block writes dz and the loss, one launch
        if (dtype == TNN_F32)
            hipLaunchKernelGGL((nll_fwd_bwd_kernel<float>), 1, kThreads, 0, s, (const float*)z, (const float*)y, m, c,
                               inv_m, (const float*)stats, (double*)nullptr, (float*)dz, (float*)loss_out);
        else
            hipLaunchKernelGGL((nll_fwd_bwd_kernel<double>), 1, kThreads, 0, s, (const double*)z, (const double*)y, m,
                               c, inv_m, (const double*)stats, (double*)nullptr, (double*)dz, (double*)loss_out);
        TNN_LAUNCH_OK();
        return 0;
    }
    void* ws = nullptr;
    if (tnn_malloc((size_t)nb * sizeof(double), &ws)) return 1;
    if (dtype == TNN_F32) {
        hipLaunchKernelGGL((nll_fwd_bwd_kernel<float>), (unsigned)nb, kThreads, 0, s, (const float*)z,
                           (const float*)y, m, c, inv_m, (const float*)stats, (double*)ws, (float*)dz, (float*)nullptr);
        if (loss_out)
            hipLaunchKernelGGL((sum_partials_kernel<float>), 1, 64, 0, s, (const double*)ws, (int)nb, (float*)loss_out);
    } else {
        hipLaunchKernelGGL((nll_fwd_bwd_kernel<double>), (unsigned)nb, kThreads, 0, s, (const double*)z,
                           (const double*)y, m, c, inv_m, (const double*)stats, (double*)ws, (double*)dz, (double*)nullptr);
        if (loss_out)
            hipLaunchKernelGGL((sum_partials_kernel<double>), 1, 64, 0, s, (const double*)ws, (int)nb, (double*)loss_out);
    }
    tnn_free(ws);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_softmax_nll_fused(const void* z, const void* y, int64_t m, int64_t c, void* stats_out,
                          void* loss_out, void* dz, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(dtype == TNN_F32 || dtype == TNN_F64, "tnn_softmax_nll_fused: dtype %d", dtype);
    TNN_REQUIRE(m > 0 && c > 0, "tnn_softmax_nll_fused: empty batch");
    // classifier heads (c <= 16) take the one-thread-per-row kernel up to 1024 rows whatever m * c is; everything else must
    // fit the LDS image of the element-parallel kernel
    const bool rows_form = c <= 16 && m <= kNllMaxRows;
    if (!rows_form && (m * c > (dtype == TNN_F32 ? NllCap<float>::elems : NllCap<double>::elems) || m > kNllMaxRows)) {   // too big for one block: the multi-block sequence
        void* st = stats_out;
        void* tmp = nullptr;
        if (!st) {
            if (tnn_malloc(16, &tmp)) return 1;
            st = tmp;
        }
        int rc = tnn_softmax_nll_stats(z, m, c, st, dtype);
        if (!rc) rc = tnn_softmax_nll_fwd_bwd(z, y, m, c, m, st, loss_out, dz, dtype);
        if (tmp) tnn_free(tmp);
        return rc;
    }
    return tnn_softmax_nll_fused_tick(z, y, m, c, m, 0, stats_out, loss_out, dz, dtype, nullptr, 0.0, 0.0);
}

int tnn_softmax_nll_fused_sharded(const void* z, const void* y, int64_t m, int64_t c, int64_t m_global,
                                  void* stats_out, void* loss_out, void* dz, int dtype) {
    return tnn_softmax_nll_fused_tick(z, y, m, c, m_global, 1, stats_out, loss_out, dz, dtype, nullptr, 0.0, 0.0);
}

int tnn_softmax_nll_fused_tick(const void* z, const void* y, int64_t m, int64_t c, int64_t m_global, int sharded,
                               void* stats_out, void* loss_out, void* dz, int dtype, void* adam_pows_f64, double b1,
                               double b2) {
    TNN_NEED_INIT();
    TNN_REQUIRE(dtype == TNN_F32 || dtype == TNN_F64, "tnn_softmax_nll_fused_tick: dtype %d", dtype);
    TNN_REQUIRE(m > 0 && c > 0, "tnn_softmax_nll_fused_tick: empty batch");
    TNN_REQUIRE((c <= 16 && m <= kNllMaxRows) ||
                    (m * c <= (dtype == TNN_F32 ? NllCap<float>::elems : NllCap<double>::elems) && m <= kNllMaxRows),
                "tnn_softmax_nll_fused_tick: %lld x %lld does not fit one workgroup", (long long)m, (long long)c);
    double* tick = (double*)adam_pows_f64;
    // classifier heads: one thread per row, one block reduction (nll_rows_kernel)
    const bool rows_kernel = c <= 16 && m <= 1024;
    const int row_threads = (int)((m + 63) / 64 * 64);
    if (sharded) {
        TNN_REQUIRE(dtype == TNN_F32, "tnn_softmax_nll_fused_tick: the sharded form is f32 only (dtype %d)", dtype);
        TNN_REQUIRE(m_global >= m, "tnn_softmax_nll_fused_tick: m_global < m");
        tnn::p2p::LaunchCtx ctx;
        if (int rc = tnn::p2p_refuse_if_failed("tnn_softmax_nll_fused_tick")) return rc;
        TNN_REQUIRE(tnn::p2p_launch_ctx(&ctx), "tnn_softmax_nll_fused_tick: the peer-to-peer transport is not enabled");
        if (rows_kernel)
            hipLaunchKernelGGL((nll_rows_kernel<float, true>), 1, row_threads, 0, tnn::stream(), (const float*)z,
                               (const float*)y, (int)m, (int)c, (float*)stats_out, (float*)loss_out, (float*)dz,
                               1.0 / (double)m_global, ctx, tick, b1, b2);
        else
            hipLaunchKernelGGL((nll_fused_kernel<float, true>), 1, 1024, 0, tnn::stream(), (const float*)z,
                               (const float*)y, (int)m, (int)c, (float*)stats_out, (float*)loss_out, (float*)dz,
                               1.0 / (double)m_global, ctx, tick, b1, b2);
    } else {
        TNN_REQUIRE(m_global == m, "tnn_softmax_nll_fused_tick: unsharded call with m_global != m");
        const int threads = m * c >= 512 ? 1024 : 256;
        if (rows_kernel && dtype == TNN_F32)
            hipLaunchKernelGGL((nll_rows_kernel<float, false>), 1, row_threads, 0, tnn::stream(), (const float*)z,
                               (const float*)y, (int)m, (int)c, (float*)stats_out, (float*)loss_out, (float*)dz, 0.0,
                               tnn::p2p::LaunchCtx{}, tick, b1, b2);
        else if (rows_kernel)
            hipLaunchKernelGGL((nll_rows_kernel<double, false>), 1, row_threads, 0, tnn::stream(), (const double*)z,
                               (const double*)y, (int)m, (int)c, (double*)stats_out, (double*)loss_out, (double*)dz,
                               0.0, tnn::p2p::LaunchCtx{}, tick, b1, b2);
        else if (dtype == TNN_F32)
            hipLaunchKernelGGL((nll_fused_kernel<float, false>), 1, threads, 0, tnn::stream(), (const float*)z,
                               (const float*)y, (int)m, (int)c, (float*)stats_out, (float*)loss_out, (float*)dz, 0.0,
                               tnn::p2p::LaunchCtx{}, tick, b1, b2);
        else
            hipLaunchKernelGGL((nll_fused_kernel<double, false>), 1, threads, 0, tnn::stream(), (const double*)z,
                               (const double*)y, (int)m, (int)c, (double*)stats_out, (double*)loss_out, (double*)dz,
                               0.0, tnn::p2p::LaunchCtx{}, tick, b1, b2);
    }
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_mse_fwd_bwd(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out,
                    void* dpred, int dtype) {
    return tnn_mse_fwd_bwd_tick(pred, y, n, m_global, loss_out, nullptr, dpred, dtype, nullptr, 1.0, 1.0);
}

int tnn_mse_fwd_bwd_tick(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out, void* loss_out2,
                         void* dpred, int dtype, void* adam_pows_f64, double b1, double b2) {
    TNN_NEED_INIT();
    TNN_REQUIRE(loss_out != nullptr || loss_out2 == nullptr, "tnn_mse_fwd_bwd_tick: loss_out2 needs loss_out");
    TNN_REQUIRE(dtype == TNN_F32 || dtype == TNN_F64, "tnn_mse_fwd_bwd: dtype %d", dtype);
    TNN_REQUIRE(n > 0 && m_global > 0, "tnn_mse_fwd_bwd: empty batch");
    int64_t nb = tnn::stream_grid(n, kThreads);
    if (nb > 1024) nb = 1024;
    void* ws = nullptr;
    if (tnn_malloc((size_t)nb * sizeof(double), &ws)) return 1;
    hipStream_t s = tnn::stream();
    double inv_m = 1.0 / (double)m_global;
    if (dtype == TNN_F32) {
        hipLaunchKernelGGL((mse_fwd_bwd_kernel<float>), (unsigned)nb, kThreads, 0, s, (const float*)pred,
                           (const float*)y, n, inv_m, (double*)ws, (float*)dpred, (double*)adam_pows_f64, b1, b2,
                           tnn::update_guard());
        if (loss_out)
            hipLaunchKernelGGL((sum_partials_kernel<float>), 1, 64, 0, s, (const double*)ws, (int)nb, (float*)loss_out,
                               (float*)loss_out2);
    } else {
        hipLaunchKernelGGL((mse_fwd_bwd_kernel<double>), (unsigned)nb, kThreads, 0, s, (const double*)pred,
                           (const double*)y, n, inv_m, (double*)ws, (double*)dpred, (double*)adam_pows_f64, b1, b2,
                           tnn::update_guard());
        if (loss_out)
            hipLaunchKernelGGL((sum_partials_kernel<double>), 1, 64, 0, s, (const double*)ws, (int)nb, (double*)loss_out,
                               (double*)loss_out2);
    }
    tnn_free(ws);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_sgd(void* p, const void* g, int64_t n, double lr, int dtype) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    unsigned grid = tnn::stream_grid(n, kThreads);
    switch (dtype) {
        case TNN_F32: hipLaunchKernelGGL((sgd_kernel<float>), grid, kThreads, 0, tnn::stream(), (float*)p, (const float*)g, n, (float)lr, tnn::update_guard()); break;
        case TNN_F64: hipLaunchKernelGGL((sgd_kernel<double>), grid, kThreads, 0, tnn::stream(), (double*)p, (const double*)g, n, lr, tnn::update_guard()); break;
        default: tnn::set_error("tnn_sgd: dtype %d is not a float type", dtype); return 2;
    }
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_optim_step(int kind, void* p, const void* g, void* s1, void* s2, void* step_out, int64_t n, double lr,
                   double a, double b, double eps, int dtype) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    TNN_REQUIRE(kind >= TNN_OPT_MOMENTUM && kind <= TNN_OPT_ADADELTA, "tnn_optim_step: unknown optimizer %d", kind);
    TNN_REQUIRE(dtype == TNN_F32 || dtype == TNN_F64, "tnn_optim_step: dtype %d is not a float type", dtype);
    TNN_REQUIRE(g && s1 && (p || step_out), "tnn_optim_step: g, s1 and one of p / step_out are required");
    TNN_REQUIRE(s2 || kind == TNN_OPT_MOMENTUM || kind == TNN_OPT_ADAGRAD, "tnn_optim_step: this optimizer needs s2");
    const unsigned grid = tnn::stream_grid(n, kThreads);
    hipStream_t st = tnn::stream();
#define TNN_OPT_LAUNCH(T, K)                                                                                      \
    hipLaunchKernelGGL((optim_kernel<T, K>), grid, kThreads, 0, st, (T*)p, (const T*)g, (T*)s1, (T*)s2, (T*)step_out, \
                       n, (T)lr, (T)a, (T)b, (T)eps, tnn::update_guard())
#define TNN_OPT_KINDS(T)                                             \
    switch (kind) {                                                  \
        case TNN_OPT_MOMENTUM: TNN_OPT_LAUNCH(T, TNN_OPT_MOMENTUM); break; \
        case TNN_OPT_RMSPROP: TNN_OPT_LAUNCH(T, TNN_OPT_RMSPROP); break;   \
        case TNN_OPT_ADAGRAD: TNN_OPT_LAUNCH(T, TNN_OPT_ADAGRAD); break;   \
        default: TNN_OPT_LAUNCH(T, TNN_OPT_ADADELTA); break;               \
    }
    if (dtype == TNN_F32) { TNN_OPT_KINDS(float) } else { TNN_OPT_KINDS(double) }
#undef TNN_OPT_KINDS
#undef TNN_OPT_LAUNCH
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_adam(void* p, const void* g, void* m, void* v, int64_t n, double lr, double b1, double b2,
             double eps, void* pows_f64, void* step_out, int dtype) {
    return tnn_adam_ex(p, g, m, v, n, lr, b1, b2, eps, pows_f64, step_out, dtype, 1, nullptr, nullptr);
}

int tnn_adam_ex(void* p, const void* g, void* m, void* v, int64_t n, double lr, double b1, double b2,
                double eps, void* pows_f64, void* step_out, int dtype, int advance, const void* scalar_src,
                void* scalar_dst) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    TNN_REQUIRE(pows_f64 != nullptr, "tnn_adam: pows state is NULL");
    TNN_REQUIRE((scalar_src == nullptr) == (scalar_dst == nullptr), "tnn_adam_ex: scalar_src / scalar_dst go together");
    hipStream_t s = tnn::stream();
    const bool vec32 = dtype == TNN_F32 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                                             reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                                             reinterpret_cast<uintptr_t>(step_out)) & 15) == 0;
    // small arenas (the MNIST net's 235 k parameters): the update launch advances the beta powers itself (see adam_kernel)
    if (advance && vec32 && n <= (int64_t(1) << 20)) {
        unsigned grid = tnn::stream_grid((n + 3) / 4, kThreads);
        if (grid > 128) grid = 128;
        hipLaunchKernelGGL((adam_kernel<float, 4>), grid, kThreads, 0, s, (float*)p, (const float*)g, (float*)m, (float*)v, n,
                           (float)lr, (float)b1, (float)b2, (float)eps, (const double*)pows_f64, (float*)step_out,
                           tnn::update_guard(), (const float*)scalar_src, (float*)scalar_dst, (double*)pows_f64, b1, b2);
        TNN_LAUNCH_OK();
        return 0;
    }
    if (advance) {         // the prologue thread that advances the beta powers also carries the scalar
        hipLaunchKernelGGL(adam_advance_kernel, 1, 1, 0, s, (double*)pows_f64, b1, b2, scalar_src, scalar_dst,
                           dtype == TNN_F64 ? 8 : 4, tnn::update_guard());
        scalar_src = nullptr;
        scalar_dst = nullptr;
    }
    if (dtype == TNN_F32) {
        bool vec = vec32;
        unsigned grid = tnn::stream_grid(vec ? (n + 3) / 4 : n, kThreads);
        if (vec)
            hipLaunchKernelGGL((adam_kernel<float, 4>), grid, kThreads, 0, s, (float*)p, (const float*)g,
                               (float*)m, (float*)v, n, (float)lr, (float)b1, (float)b2, (float)eps,
                               (const double*)pows_f64, (float*)step_out, tnn::update_guard(), (const float*)scalar_src,
                               (float*)scalar_dst);
        else
            hipLaunchKernelGGL((adam_kernel<float, 1>), grid, kThreads, 0, s, (float*)p, (const float*)g,
                               (float*)m, (float*)v, n, (float)lr, (float)b1, (float)b2, (float)eps,
                               (const double*)pows_f64, (float*)step_out, tnn::update_guard(), (const float*)scalar_src,
                               (float*)scalar_dst);
    } else if (dtype == TNN_F64) {
        unsigned grid = tnn::stream_grid(n, kThreads);
        hipLaunchKernelGGL((adam_kernel<double, 1>), grid, kThreads, 0, s, (double*)p, (const double*)g,
                           (double*)m, (double*)v, n, lr, b1, b2, eps, (const double*)pows_f64,
                           (double*)step_out, tnn::update_guard(), (const double*)scalar_src, (double*)scalar_dst);
    } else {
        tnn::set_error("tnn_adam: dtype %d is not a float type", dtype);
        return 2;
    }
    TNN_LAUNCH_OK();
    return 0;
}

}  // extern "C"
