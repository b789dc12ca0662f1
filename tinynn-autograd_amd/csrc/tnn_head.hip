// Fused classifier head (K8 + K9 for the MNIST-size step): everything that touches the narrow last
// Dense layer of the step in ONE launch, because each piece is far below a microsecond of math and would
// otherwise pay a kernel boundary each:
//     z  = a W + b                                  core/layers.py:49        (forward of the last Dense)
//     M, S, loss, dz = whole-batch softmax NLL      core/losses.py:24-32
//     dW = a^T dz,  db = column-sum dz              core/ops.py:159-160, :52-54
//     da = (dz W^T) * [pre-activation >= 0]         core/ops.py:156-157, :342-343 (mask = sign bit of a)
// MEASURED (MI355X): 14.7 us as one launch vs 2.4 + 5.3 + 2.8 us for the three-launch sequence below it —
// a single workgroup pays every phase's LDS / barrier / load latency serially on one CU, while multi-block
// kernels of this size cost only ~0.3-1 us over the 1.6 us launch floor.  The fused form is therefore OFF by
// default (TNN_HEAD_FUSION=1 enables it; it stays parity-tested) and the entry point runs the sequence.
// One 1024-thread workgroup on one CU; a, W, z/e/dz live in LDS (dynamic, up to ~120 KiB of the 160 KiB);
// the three small GEMMs run on v_mfma_f32_16x16x4_f32 out of LDS, the softmax part is element-parallel.
//   a  : [m, H]  H % 16 == 0 (rows padded to 16 in LDS with zeros)      W : [H, C], C <= 16 (padded to 16)
#include <math.h>
#include <stdlib.h>

#include "tnn_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 1024;
constexpr int kCP = 16;   // padded class count

template <bool IS_MAX>
__device__ __forceinline__ double block_reduce(double v, double* red, double* bcast) {
    v = IS_MAX ? tnn::wave_max(v) : tnn::wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (w == 0) {
        double r = lane < (kThreads / 64) ? red[lane] : (IS_MAX ? -INFINITY : 0.0);
        r = IS_MAX ? tnn::wave_max(r) : tnn::wave_sum(r);
        if (lane == 0) *bcast = r;
    }
    __syncthreads();
    return *bcast;
}

struct HeadArgs {
    int m, H, C;
    const float *a, *w, *b, *y;
    float *logits, *dz, *stats, *loss, *dw, *db, *da;
};

__global__ __launch_bounds__(kThreads) void mlp_head_kernel(HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = p.m, H = p.H, C = p.C;
    const int mp = (m + 15) & ~15, SA = H + 4;          // padded rows; a row stride (floats), conflict-free b128
    float* a_s = reinterpret_cast<float*>(smem);         // [mp][SA]
    float* w_s = a_s + (size_t)mp * SA;                  // [H][16]
    float* z_s = w_s + (size_t)H * kCP;                  // [mp][16]  logits -> exp -> dz (in place)
    float* y_s = z_s + (size_t)mp * kCP;                 // [mp][16]  labels -> e * y
    double* q_s = reinterpret_cast<double*>(y_s + (size_t)mp * kCP);   // [mp]
    double* red = q_s + mp;                              // [16] wave partials + [4] block scalars
    double* bcast = red + 20;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i16 = lane & 15, grp = lane >> 4;

    // ---- stage a (float4, coalesced), W (zero-padded to 16 columns)
    const int hv = H / 4;
    for (int f = tid; f < mp * hv; f += kThreads) {
        const int r = f / hv, c4 = f - r * hv;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < m) v = *reinterpret_cast<const float4*>(p.a + (size_t)r * H + c4 * 4);
        *reinterpret_cast<float4*>(a_s + (size_t)r * SA + c4 * 4) = v;
    }
    for (int f = tid; f < H * kCP; f += kThreads) {
        const int h = f >> 4, c = f & 15;
        w_s[f] = c < C ? p.w[(size_t)h * C + c] : 0.f;
    }
    for (int f = tid; f < mp * kCP; f += kThreads) {          // labels too: no global reads after this point
        const int r = f >> 4, c = f & 15;
        y_s[f] = (r < m && c < C) ? p.y[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();

    // ---- z = a W + b : one 16-row tile per wave iteration, K = H
    const int row_tiles = mp / 16;
    for (int t = wid; t < row_tiles; t += kThreads / 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // two chains hide the 40-cycle MFMA latency
        const float* arow = a_s + (size_t)(t * 16 + i16) * SA + grp * 4;
#pragma unroll 4
        for (int k = 0; k < H; k += 16) {
            const float4 av = *reinterpret_cast<const float4*>(arow + k);
            const float* wp = w_s + (size_t)(k + grp * 4) * kCP + i16;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wp[0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wp[kCP], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wp[2 * kCP], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wp[3 * kCP], acc2, 0, 0, 0);
        }
        acc += acc2;
        const float bias = i16 < C ? p.b[i16] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = t * 16 + grp * 4 + r;
            const float v = acc[r] + bias;
            z_s[row * kCP + i16] = v;
            if (p.logits && row < m && i16 < C) p.logits[(size_t)row * C + i16] = v;
        }
    }
    __syncthreads();

    // ---- whole-batch softmax NLL, element-parallel over the padded [mp,16] grid
    const int n16 = mp * kCP;
    double mx = -INFINITY;
    for (int f = tid; f < n16; f += kThreads) {
        const int r = f >> 4, c = f & 15;
        if (r < m && c < C) { double v = (double)z_s[f]; mx = v > mx ? v : mx; }
    }
    const double M = block_reduce<true>(mx, red, bcast);
    const float Mf = (float)M;
    double s = 0.0;
    for (int f = tid; f < n16; f += kThreads) {
        const int r = f >> 4, c = f & 15;
        float e = 0.f;
        if (r < m && c < C) { e = expf(z_s[f] - Mf); s += (double)e; }   // f32 exp, f64 accumulation
        z_s[f] = e;                                            // exp values replace the logits
        y_s[f] = e * y_s[f];                                   // e * y (zero in the padding)
    }
    const double S = block_reduce<false>(s, red, bcast);
    if (tid == 0) { red[18] = log(S); red[19] = 1.0 / S; }    // scalars once per block (slots beyond the 16 waves' use)
    const double inv_m = 1.0 / (double)m;
    double local = 0.0;
    for (int r = tid; r < mp; r += kThreads) {
        double qinv = 0.0;
        if (r < m) {
            double q = 0.0;
#pragma unroll
            for (int c = 0; c < kCP; ++c) q += (double)y_s[r * kCP + c];
            local -= (double)logf((float)q);
            qinv = inv_m / q;
        }
        q_s[r] = qinv;
    }
    __syncthreads();
    const double log_s = red[18];
    const float inv_s = (float)red[19];
    const double loss = log_s + block_reduce<false>(local, red, bcast) * inv_m;   // log S - mean(log q)
    for (int f = tid; f < n16; f += kThreads) {
        const float d = z_s[f] * inv_s - y_s[f] * (float)q_s[f >> 4];               // 0 in the padding
        const int r = f >> 4, c = f & 15;
        if (p.dz && r < m && c < C) p.dz[(size_t)r * C + c] = d;
        z_s[f] = d;                                            // dz replaces the exp values
    }
    if (tid == 0) {
        if (p.loss) p.loss[0] = (float)loss;
        if (p.stats) { p.stats[0] = (float)M; p.stats[1] = (float)S; }
    }
    __syncthreads();

    // ---- dW = a^T dz  (M = H, N = 16, K = rows), db = column sums of dz
    const int h_tiles = H / 16;
    for (int t = wid; t < h_tiles; t += kThreads / 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < mp; k += 16) {
            const float* ap = a_s + (size_t)(k + grp * 4) * SA + t * 16 + i16;     // a[k+4g+j][h0+i]
            const float* dp = z_s + (size_t)(k + grp * 4) * kCP + i16;             // dz[k+4g+j][c=i]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], dp[0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[SA], dp[kCP], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[2 * SA], dp[2 * kCP], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[3 * SA], dp[3 * kCP], acc2, 0, 0, 0);
        }
        acc += acc2;
        if (i16 < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p.dw[(size_t)(t * 16 + grp * 4 + r) * C + i16] = acc[r];
        }
    }
    if (wid == kThreads / 64 - 1) {                            // last wave: db[c] = sum_r dz[r][c]
        float sum = 0.f;
        for (int r = grp; r < mp; r += 4) sum += z_s[r * kCP + i16];
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (lane < C) p.db[lane] = sum;
    }

    // ---- da = (dz W^T) * mask(a)  (M = rows, N = H, K = 16 padded classes); only LDS reads, no barrier needed
    if (p.da) {
        const int kc = (C + 3) / 4;
        for (int t = wid; t < row_tiles * h_tiles; t += kThreads / 64) {
            const int rt = t / h_tiles, ht = t - rt * h_tiles;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* dp = z_s + (size_t)(rt * 16 + i16) * kCP + grp;           // dz[r0+i][4s+g]
            const float* wp = w_s + (size_t)(ht * 16 + i16) * kCP + grp;           // W[h0+i][4s+g]
            for (int sidx = 0; sidx < kc; ++sidx)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[4 * sidx], wp[4 * sidx], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + grp * 4 + r, col = ht * 16 + i16;
                if (row < m) {
                    const float av = a_s[(size_t)row * SA + col];
                    p.da[(size_t)row * H + col] = (__float_as_uint(av) >> 31) ? 0.f : acc[r];
                }
            }
        }
    }
}

size_t head_lds_bytes(int64_t m, int64_t H) {
    int64_t mp = (m + 15) & ~int64_t(15);
    return (size_t)(mp * (H + 4) + H * kCP + 2 * mp * kCP) * 4 + (size_t)(mp + 20 + 2) * 8;
}

bool g_attr_set = false;

}  // namespace

extern "C" {

int tnn_mlp_head(int64_t rows, int64_t n_hidden, int64_t n_classes, const void* a, const void* w,
                 const void* b, const void* y, void* logits, void* dz, void* stats, void* loss, void* dw,
                 void* db, void* da, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(rows > 0 && n_hidden > 0 && n_classes > 0, "tnn_mlp_head: empty head");
    TNN_REQUIRE(logits != nullptr && dz != nullptr && dw != nullptr && db != nullptr,
                "tnn_mlp_head: logits, dz, dw and db buffers are required");
    const bool aligned = ((reinterpret_cast<uintptr_t>(a)) & 15) == 0;
    const bool fused = dtype == TNN_F32 && n_classes <= kCP && n_hidden % 16 == 0 && aligned &&
                       head_lds_bytes(rows, n_hidden) <= 120 * 1024 && getenv("TNN_HEAD_FUSION") != nullptr;
    if (fused) {
        size_t lds = head_lds_bytes(rows, n_hidden);
        if (!g_attr_set) {
            TNN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_head_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            g_attr_set = true;
        }
        HeadArgs p;
        p.m = (int)rows; p.H = (int)n_hidden; p.C = (int)n_classes;
        p.a = (const float*)a; p.w = (const float*)w; p.b = (const float*)b; p.y = (const float*)y;
        p.logits = (float*)logits; p.dz = (float*)dz; p.stats = (float*)stats; p.loss = (float*)loss;
        p.dw = (float*)dw; p.db = (float*)db; p.da = (float*)da;
        hipLaunchKernelGGL(mlp_head_kernel, 1, kThreads, lds, tnn::stream(), p);
        TNN_LAUNCH_OK();
        return 0;
    }
    // general shapes: the same maths as three launches
    if (int rc = tnn_gemm_bias_act(0, 0, rows, n_classes, n_hidden, a, n_hidden, w, n_classes, b, TNN_ACT_NONE, 0,
                                   logits, n_classes, dtype))
        return rc;
    if (int rc = tnn_softmax_nll_fused(logits, y, rows, n_classes, stats, loss, dz, dtype)) return rc;
    return tnn_dense_bwd(rows, n_hidden, n_classes, a, dz, w, dw, db, da, a, dtype);
}

}  // extern "C"
