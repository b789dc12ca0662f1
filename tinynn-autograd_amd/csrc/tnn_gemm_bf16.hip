// K1 (bf16 variant, BASELINE.json configs[4]): C = A * B^T with bf16 operands and fp32 accumulation on
// v_mfma_f32_32x32x16_bf16.  ONE layout is implemented — both operands K-contiguous (A stored [M,K], B stored
// [N,K]) — because the bf16 trainer keeps every operand in that form:
//     forward  z  = a  W        ->  A = a [m,in],        B = W^T [out,in]   (transposed weight copy, refreshed
//                                                                            after every optimizer update)
//     dX       da = dz W^T      ->  A = dz [m,out],      B = W [in,out]
//     dW       dW = a^T dz      ->  A = a^T [in,m],      B = dz^T [out,m]   (transposed activation copies made by
//                                                                            the LDS-tiled bf16 transpose below)
// so no fragment ever needs a transposing LDS read.  Tile 128x128x64, 4 waves of 64x64 (2x2 MFMA tiles), LDS rows
// padded to 144 B (conflict-free ds_read_b128, same argument as the fp32 kernel), and the fp32 kernel's software
// pipeline: fragment slots filled two 16-deep chunks ahead, next tile's LDS store + global loads of tile kt+2 in
// chunk 0, one barrier per K-tile after chunk 1.  Epilogues: plain (f32 or bf16 out), bias + ReLU with the mask
// in the sign bit of zero (bf16 out), or multiply by the mask of a previous ReLU output (bf16 out).
// Shape contract of the fast path: K % 64 == 0, lda/ldb % 8 == 0, 16-B aligned bases; M, N arbitrary (rows
// beyond the edge read a clamped address and are never stored).
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "tnn_internal.h"

namespace {

#include "tnn_gemm_bf16_types.h"

constexpr int BM = 128, BN = 128, BK = 64, WM = 2, WN = 2, NT = 256;
constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
constexpr int SROW = BK + 8;                 // LDS row stride in bf16 elements (144 B)
constexpr int A_ELEMS = BM * SROW, B_ELEMS = BN * SROW;
constexpr int A_V = BM * BK / 8 / NT;        // 16-B vectors per thread per tile (4)
constexpr int B_V = BN * BK / 8 / NT;
constexpr int KK = BK / 16;                  // MFMA k-steps per tile (4)

__global__ __launch_bounds__(NT) void gemm_bf16_nt_kernel(BfArgs g) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (A_ELEMS + B_ELEMS)];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int nb = g.tiles_m * g.tiles_n;
    const int t = xcd_remap16((int)blockIdx.x, nb);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * g.tiles_n;
    const int first_m = (t / per_group) * GROUP_M;
    const int gsz = min(g.tiles_m - first_m, GROUP_M);
    const int64_t m0 = (int64_t)(first_m + (t % per_group) % gsz) * BM;
    const int64_t n0 = (int64_t)((t % per_group) / gsz) * BN;
    const int nk = (int)(g.K / BK);

    // staging geometry: vector f -> row f / 8, 16-B column f % 8
    uint32_t a_off[A_V], b_off[B_V];      // constant per-thread byte offsets (operands below 4 GiB, checked on the host)
    int a_dst[A_V], b_dst[B_V];
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
        const int f = tid + i * NT, row = f / (BK / 8), c8 = f % (BK / 8);
        const int64_t gm = m0 + row;
        a_off[i] = (uint32_t)(((gm < g.M ? gm : 0) * g.lda + c8 * 8) * 2);
        a_dst[i] = row * SROW + c8 * 8;
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
        const int f = tid + i * NT, row = f / (BK / 8), c8 = f % (BK / 8);
        const int64_t gn = n0 + row;
        b_off[i] = (uint32_t)(((gn < g.N ? gn : 0) * g.ldb + c8 * 8) * 2);
        b_dst[i] = row * SROW + c8 * 8;
    }
    u32x4 ra[A_V], rb[B_V];
    // buffer loads: SGPR resource + constant VGPR offset + scalar K offset (no vector address arithmetic in the loop)
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(g.A), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(g.B), 0, 0xffffffffu, 0x00020000);
    auto load_tile = [&](int kt) {
        const uint32_t koff = (uint32_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < A_V; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_off[i], koff, 0);
#pragma unroll
        for (int i = 0; i < B_V; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_off[i], koff, 0);
    };
    auto store_tile = [&](int buf) {
        bf16_t* As = lds + buf * (A_ELEMS + B_ELEMS);
        bf16_t* Bs = As + A_ELEMS;
#pragma unroll
        for (int i = 0; i < A_V; ++i) *reinterpret_cast<u32x4*>(As + a_dst[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_V; ++i) *reinterpret_cast<u32x4*>(Bs + b_dst[i]) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment: lane (row l31, k-group lhi) holds 8 consecutive k: elements [kk*16 + lhi*8, +8)
    const int a_frag = (wm * TM + l31) * SROW + lhi * 8;
    const int b_frag = (wn * TN + l31) * SROW + lhi * 8;
    bf16x8 af[KK][MI], bfr[KK][NI];
    auto read_frag = [&](int buf, int kk) {
        const bf16_t* As = lds + buf * (A_ELEMS + B_ELEMS);
        const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            af[kk][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + a_frag + i * 32 * SROW + kk * 16));
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            bfr[kk][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bs + b_frag + i * 32 * SROW + kk * 16));
        }
    };
    auto mfma_chunk = [&](int kk) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][mi], bfr[kk][ni], acc[mi][ni], 0, 0, 0);
    };

    static_assert(KK == 4, "pipeline written for four 16-deep chunks per K-tile");
    if (nk > 0) {
        load_tile(0);
        store_tile(0);
        if (nk > 1) load_tile(1);
    }
    __syncthreads();
    if (nk > 0) {
        read_frag(0, 0);
        read_frag(0, 1);
    }
    constexpr int NMF = MI * NI;
    // the barrier with the MFMAs of chunks 2-3 tied behind it (see tnn_gemm.hip: hipcc otherwise hoists them above
    // the barrier and the post-barrier fragment reads have nothing to hide behind)
    auto pinned_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int kk = 2; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(af[kk][i]));
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(bfr[kk][i]));
        }
    };
#define TNN_BF16_TILE(CUR, KT)                                                 \
    store_tile((CUR) ^ 1);                                                     \
    load_tile((KT) + 2);                                                       \
    read_frag((CUR), 2);                                                       \
    read_frag((CUR), 3);                                                       \
    mfma_chunk(0);                                                             \
    mfma_chunk(1);                                                             \
    _Pragma("unroll") for (int q = 0; q < 2 * NMF; ++q) {                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
    pinned_barrier();                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
    read_frag((CUR) ^ 1, 0);                                                   \
    read_frag((CUR) ^ 1, 1);                                                   \
    mfma_chunk(2);                                                             \
    mfma_chunk(3);                                                             \
    _Pragma("unroll") for (int q = 0; q < 2 * NMF; ++q) {                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);
    int kt = 0;
    for (; kt + 1 < nk - 2; kt += 2) {              // steady state: tiles kt+1 and kt+2 exist, no branches
        TNN_BF16_TILE(0, kt)
        TNN_BF16_TILE(1, kt + 1)
    }
#undef TNN_BF16_TILE
    for (; kt < nk; ++kt) {                         // last two or three tiles
        const int cur = kt & 1;
        const bool has1 = kt + 1 < nk;
        if (has1) store_tile(cur ^ 1);
        if (kt + 2 < nk) load_tile(kt + 2);
        read_frag(cur, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(0);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(cur, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (has1) read_frag(cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(2);
        __builtin_amdgcn_sched_barrier(0);
        if (has1) read_frag(cur ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(3);
        __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue (32x32 C/D layout: col = l31, row = (r&3) + 8*(r>>2) + 4*lhi)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int64_t col = n0 + wn * TN + ni * 32 + l31;
            if (col >= g.N) continue;
            const float bias = (g.epi == BEPI_BIAS_ACT && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= g.M) continue;
                float v = acc[mi][ni][r];
                if (g.epi == BEPI_BIAS_ACT) {
                    v += bias;
                    if (g.act == TNN_ACT_RELU) v = v < 0.f ? (g.relu_sign ? -0.0f : 0.f) : fabsf(v);
                } else if (g.epi == BEPI_MASK) {
                    if (g.Y[row * g.ldy + col] & 0x8000u) v = 0.f;
                }
                if (g.c_bf16) reinterpret_cast<bf16_t*>(g.C)[row * g.ldc + col] = f2bf(v);
                else reinterpret_cast<float*>(g.C)[row * g.ldc + col] = v;
            }
        }
}

#include "tnn_gemm_bf16_dma.h"     // gemm_bf16_dma_kernel<NW, NS>: the LDS-DMA variant (see its header)
#include "tnn_gemm_bf16_sk.h"      // sk::gemm_bf16_sk_kernel: 256-row tiles + split-K for the skinny (M = 512) products

// Hand-off memory of the split-K kernel: fp32 slabs [tile][slice][256 x 128] + two counter words per tile, one set per stream
// (two such GEMMs on different streams must not share counters).  Allocated ONCE per stream, on first use outside a capture,
// at its upper bound — sk_shape() admits at most num_cus (tile, slice) pairs, i.e. 32 MB of slabs on 256 CUs — and never
// returned or regrown: the pointers are kernel arguments of every hipGraph captured since, so they must stay valid whatever
// shapes later trainers ask for (growing by hipFree + hipMalloc left earlier captures replaying into freed memory).  The
// counters are zeroed once — every launch leaves them at zero.
struct SkWorkspace {
    float* slabs = nullptr;
    unsigned* counters = nullptr;
    size_t slab_bytes = 0;
    int tiles = 0;
};
std::mutex g_sk_mu;
std::unordered_map<hipStream_t, SkWorkspace> g_sk_ws;

bool sk_workspace(hipStream_t s, int tiles, int slices, size_t slab_bytes_per_slice, SkWorkspace* out) {
    std::lock_guard<std::mutex> lk(g_sk_mu);
    SkWorkspace& w = g_sk_ws[s];
    const size_t need = (size_t)tiles * slices * slab_bytes_per_slice;
    constexpr int n_cnt = 4096;
    if (w.slabs != nullptr) {                       // fixed size: a request beyond it is refused, never served by regrowing
        if (w.slab_bytes < need || w.tiles < tiles) return false;
        *out = w;
        return true;
    }
    const size_t cap = (size_t)tnn::num_cus() * slab_bytes_per_slice;      // tiles x slices <= CUs (sk_shape)
    if (need > cap || tiles > n_cnt) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;   // not inside a capture
    if (hipMalloc(&w.slabs, cap) != hipSuccess) { (void)hipGetLastError(); w = SkWorkspace(); return false; }
    if (hipMalloc(&w.counters, (size_t)n_cnt * 2 * sizeof(unsigned)) != hipSuccess ||
        hipMemset(w.counters, 0, (size_t)n_cnt * 2 * sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(w.slabs);
        w = SkWorkspace();
        return false;
    }
    w.slab_bytes = cap;
    w.tiles = n_cnt;
    *out = w;
    return true;
}

// bf16 [R, C] -> [C, R], 64x64 tiles.  2-byte accesses made the first version instruction-bound (2.2 TB/s), so:
// every thread loads a 4x4 block with four 8-B loads, transposes it in registers, writes the four transposed
// 8-B rows into an LDS image of the OUTPUT tile, and after the barrier the tile leaves with 16-B stores, eight
// lanes per 128-B output row.  Requires R % 4 == 0 and C % 4 == 0 (else the element-wise fallback below).
__device__ __forceinline__ void transpose_bf16_tile(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int64_t R, int64_t C,
                                                    const int64_t r0, const int64_t c0) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][64 + 8];     // [c][r], 144-B rows
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;             // 16 x 16 threads, 4x4 elements each
    u32x2 row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + 4 * ty + i, c = c0 + 4 * tx;
        row[i] = (r < R && c < C) ? *reinterpret_cast<const u32x2*>(in + r * C + c) : u32x2{0u, 0u};
    }
    // column j of the 4x4 block = elements {row0[j], row1[j], row2[j], row3[j]}
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int w = j >> 1;
        u32x2 col;
        if (j & 1) {
            col.x = (row[0][w] >> 16) | (row[1][w] & 0xffff0000u);
            col.y = (row[2][w] >> 16) | (row[3][w] & 0xffff0000u);
        } else {
            col.x = (row[0][w] & 0xffffu) | (row[1][w] << 16);
            col.y = (row[2][w] & 0xffffu) | (row[3][w] << 16);
        }
        *reinterpret_cast<u32x2*>(&tile[4 * tx + j][4 * ty]) = col;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (threadIdx.x >> 3) + 32 * i, seg = threadIdx.x & 7;      // output row c, 16-B segment
        const int64_t oc = c0 + c, orow = r0 + seg * 8;
        if (oc < C && orow < R) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(&tile[c][seg * 8]);
            if (orow + 8 <= R) {
                *reinterpret_cast<u32x4*>(out + oc * R + orow) = v;
            } else {                                                            // ragged edge (R % 8 == 4)
                *reinterpret_cast<u32x2*>(out + oc * R + orow) = u32x2{v.x, v.y};
            }
        }
    }
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in,
                                                             bf16_t* __restrict__ out, int64_t R, int64_t C) {
    transpose_bf16_tile(in, out, R, C, (int64_t)blockIdx.y * 64, (int64_t)blockIdx.x * 64);
}
// two independent transposes in ONE launch (tnn_transpose2_bf16: the two K-contiguous operands of a dW product, a^T and dz^T,
// written just in front of it): blocks [0, n1) work on the first matrix, the rest on the second
__global__ __launch_bounds__(256) void transpose2_bf16_kernel(const bf16_t* __restrict__ in1, bf16_t* __restrict__ out1, int64_t R1,
                                                              int64_t C1, const bf16_t* __restrict__ in2, bf16_t* __restrict__ out2,
                                                              int64_t R2, int64_t C2, int n1) {
    const int b = (int)blockIdx.x;
    if (b < n1) {                                           // block-uniform
        const int tc = (int)((C1 + 63) / 64);
        transpose_bf16_tile(in1, out1, R1, C1, (int64_t)(b / tc) * 64, (int64_t)(b % tc) * 64);
    } else {
        const int tc = (int)((C2 + 63) / 64), bb = b - n1;
        transpose_bf16_tile(in2, out2, R2, C2, (int64_t)(bb / tc) * 64, (int64_t)(bb % tc) * 64);
    }
}

__global__ __launch_bounds__(256) void transpose_bf16_slow_kernel(const bf16_t* __restrict__ in,
                                                                  bf16_t* __restrict__ out, int64_t R, int64_t C) {
    __shared__ bf16_t tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        const int64_t r = r0 + j, c = c0 + tx;
        if (r < R && c < C) tile[j][tx] = in[r * C + c];
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const int64_t c = c0 + j, r = r0 + tx;
        if (c < C && r < R) out[c * R + r] = tile[tx][j];
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                            int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f2bf(in[i]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out,
                                                            int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = bf2f(in[i]);
}

// sum((pred - y)^2) / m and dpred = 2 (pred - y) / m on bf16 tensors (f32 math, f64 block partials)
__global__ __launch_bounds__(256) void mse_bf16_kernel(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ y,
                                                       int64_t n, double inv_m, double* __restrict__ partial,
                                                       bf16_t* __restrict__ dpred, double* __restrict__ tick = nullptr,
                                                       double tb1 = 1.0, double tb2 = 1.0, const int* guard = nullptr) {
    __shared__ double lds[4];
    // Adam's {b1^t, b2^t} advanced by one thread of the step's loss launch (nothing in this launch reads them)
    if (tick != nullptr && blockIdx.x == 0 && threadIdx.x == 0 &&
        (guard == nullptr || __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)) {
        tick[0] *= tb1;
        tick[1] *= tb2;
    }
    double local = 0.0;
    const float two_inv_m = (float)(2.0 * inv_m);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float e = bf2f(pred[i]) - bf2f(y[i]);
        local += (double)e * (double)e;
        if (dpred) dpred[i] = f2bf(two_inv_m * e);
    }
    local = tnn::wave_sum(local);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds[w] = local;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((lds[0] + lds[1]) + (lds[2] + lds[3])) * inv_m;
}
// The "prep" launch of a whole bf16 training step (tnn_mse_bf16_prep), 64 x 64 tiles, two roles in ONE grid:
//   blocks [0, n_loss)     a tile of pred / y [R][C]: e = pred - y, the tile's share of sum(e^2) / m (f64 block partial),
//                          dz = 2 e / m row-major (8-B stores, the load geometry) AND dz^T [C][R] through an LDS image of the
//                          transposed tile (transpose_bf16_kernel's scheme) — the K-contiguous operand of the last layer's dW
//                          product (core/ops.py:159-160);
//                          the LAST of these blocks to finish (agent-scope arrival ticket, back at 0 afterwards) adds the block
//                          partials in index order — sum_partials_f32_kernel's order, so the loss does not depend on who is last;
//   blocks [n_loss, ...)   a tile of the batch x [R][XC] -> x^T [XC][R], the first layer's dW operand.
// Replaces mse_bf16_kernel + sum_partials_f32_kernel + two transpose_bf16_kernel launches.  R, C, XC multiples of 64.
__global__ __launch_bounds__(256) void mse_prep_bf16_kernel(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ y,
                                                            int64_t R, int64_t C, double inv_m, double* __restrict__ partial,
                                                            unsigned* __restrict__ ticket, float* __restrict__ loss_out,
                                                            float* __restrict__ loss_out2, bf16_t* __restrict__ dz,
                                                            bf16_t* __restrict__ dzT, const bf16_t* __restrict__ x, int64_t XC,
                                                            bf16_t* __restrict__ xT, double* __restrict__ tick, double tb1,
                                                            double tb2, const int* guard) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][64 + 8];     // [c][r], 144-B rows
    __shared__ double lds4[4];
    __shared__ int is_last;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;             // 16 x 16 threads, 4x4 elements each
    const int n_tc = (int)(C / 64), n_loss = (int)(R / 64) * n_tc;
    const int b = (int)blockIdx.x;
    const bool loss_role = b < n_loss;                                  // block-uniform
    if (tick != nullptr && b == 0 && threadIdx.x == 0 &&
        (guard == nullptr || __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)) {
        tick[0] *= tb1;                                                 // Adam's {b1^t, b2^t}: nothing in this launch reads them
        tick[1] *= tb2;
    }
    const int bb = loss_role ? b : b - n_loss, ntc = loss_role ? n_tc : (int)(XC / 64);
    const int64_t r0 = (int64_t)(bb / ntc) * 64, c0 = (int64_t)(bb % ntc) * 64;
    const int64_t ld = loss_role ? C : XC;
    const bf16_t* src = loss_role ? pred : x;
    bf16_t* dstT = loss_role ? dzT : xT;
    u32x2 row[4];
    double local = 0.0;
    const float two_inv_m = (float)(2.0 * inv_m);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t o = (r0 + 4 * ty + i) * ld + c0 + 4 * tx;
        row[i] = *reinterpret_cast<const u32x2*>(src + o);
        if (loss_role) {
            const u32x2 yv = *reinterpret_cast<const u32x2*>(y + o);
            float e[4] = {__uint_as_float(row[i].x << 16) - __uint_as_float(yv.x << 16),
                          __uint_as_float(row[i].x & 0xffff0000u) - __uint_as_float(yv.x & 0xffff0000u),
                          __uint_as_float(row[i].y << 16) - __uint_as_float(yv.y << 16),
                          __uint_as_float(row[i].y & 0xffff0000u) - __uint_as_float(yv.y & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 4; ++k) local += (double)e[k] * (double)e[k];
            row[i].x = (uint32_t)f2bf(two_inv_m * e[0]) | ((uint32_t)f2bf(two_inv_m * e[1]) << 16);
            row[i].y = (uint32_t)f2bf(two_inv_m * e[2]) | ((uint32_t)f2bf(two_inv_m * e[3]) << 16);
            if (dz) *reinterpret_cast<u32x2*>(dz + o) = row[i];
        }
    }
    if (dstT != nullptr) {
        // column j of the 4x4 block = elements {row0[j], row1[j], row2[j], row3[j]}
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = j >> 1;
            u32x2 col;
            if (j & 1) {
                col.x = (row[0][w] >> 16) | (row[1][w] & 0xffff0000u);
                col.y = (row[2][w] >> 16) | (row[3][w] & 0xffff0000u);
            } else {
                col.x = (row[0][w] & 0xffffu) | (row[1][w] << 16);
                col.y = (row[2][w] & 0xffffu) | (row[3][w] << 16);
            }
            *reinterpret_cast<u32x2*>(&tile[4 * tx + j][4 * ty]) = col;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (threadIdx.x >> 3) + 32 * i, seg = threadIdx.x & 7;      // output row c, 16-B segment
            *reinterpret_cast<u32x4*>(dstT + (c0 + c) * R + r0 + seg * 8) = *reinterpret_cast<const u32x4*>(&tile[c][seg * 8]);
        }
    }
    if (!loss_role) return;
    local = tnn::wave_sum(local);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds4[w] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        // system-scope (write-through) store + drain instead of a release fence: a fence at agent scope writes back every dirty
        // line of this XCD's L2 — the dz / dz^T tiles just stored — once per workgroup (34 us for the launch, measured)
        __hip_atomic_store(partial + b, ((lds4[0] + lds4[1]) + (lds4[2] + lds4[3])) * inv_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Two-level arrival: the tiles draw tickets in groups of G >= 32 (word 1 + group), the last of a group draws one of the
        // launch (word 0) — a thousand agent-scope atomics on ONE word serialise to ~25 us (measured on the data-parallel
        // step's counters, csrc/tnn_gemm.hip), 32 + 32 do not.  Every word is back at 0 when the launch ends (graph replays).
        const int G = max(32, (n_loss + 62) / 63), n_groups = (n_loss + G - 1) / G, grp = b / G;
        const unsigned members = (unsigned)min(G, n_loss - grp * G);
        int last = 0;
        if (__hip_atomic_fetch_add(ticket + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(ticket + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)n_groups - 1) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last || loss_out == nullptr || threadIdx.x >= 64) return;
    double sl = 0.0;      // (system-scope loads: this XCD's L2 may hold last launch's partials)
    for (int i = threadIdx.x; i < n_loss; i += 64) sl += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    sl = tnn::wave_sum(sl);
    if (threadIdx.x == 0) {
        loss_out[0] = (float)sl;
        if (loss_out2) loss_out2[0] = (float)sl;
    }
}
__global__ __launch_bounds__(64) void sum_partials_f32_kernel(const double* __restrict__ partial, int n,
                                                              float* __restrict__ out, float* __restrict__ out2 = nullptr) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = tnn::wave_sum(s);
    if (threadIdx.x == 0) {
        out[0] = (float)s;
        if (out2) out2[0] = (float)s;
    }
}

// Adam on the fp32 master copy + refresh of the bf16 working copy (28 B + 2 B per parameter)
// G16: the gradient itself is bf16 (the reduce-scattered slice of the sharded-optimizer step, tnn_adam_master_g16)
template <bool G16>
__global__ __launch_bounds__(256) void adam_master_bf16_kernel(float* __restrict__ p, const void* __restrict__ g_,
                                                               float* __restrict__ m, float* __restrict__ v,
                                                               bf16_t* __restrict__ w16, int64_t n, float lr, float b1,
                                                               float b2, float eps, const double* __restrict__ state,
                                                               const int* guard) {
    TNN_GUARD_RETURN(guard);
    const float* g = reinterpret_cast<const float*>(g_);
    const bf16_t* g16 = reinterpret_cast<const bf16_t*>(g_);
    const double p1 = state[0], p2 = state[1];
    const float ic1 = (float)(1.0 / (1.0 - p1)), ic2 = (float)(1.0 / (1.0 - p2));
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    // 16-B non-temporal streams for everything only the optimizer touches (g, m, v, fp32 master weights); the bf16
    // copy is re-read by the next GEMMs and takes ordinary 8-B stores.  n4 = vectorisable prefix (arenas 16-B aligned).
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | (reinterpret_cast<uintptr_t>(g_) << (G16 ? 1 : 0)) |
                           reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(w16) & 7) == 0;
    const int64_t n4 = aligned ? n / 4 : 0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < n4; i += nth) {
        f32x4 gi;
        if constexpr (G16) {
            const u32x2 gw = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(g16) + i);
            gi = f32x4{__uint_as_float(gw.x << 16), __uint_as_float(gw.x & 0xffff0000u), __uint_as_float(gw.y << 16),
                       __uint_as_float(gw.y & 0xffff0000u)};
        } else {
            gi = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        }
        f32x4 mi = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m) + i);
        f32x4 vi = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i);
        f32x4 pi = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mi[k] = mi[k] + omb1 * (gi[k] - mi[k]);
            vi[k] = vi[k] + omb2 * (gi[k] * gi[k] - vi[k]);
            pi[k] = pi[k] + (-lr * (mi[k] * ic1) / (sqrtf(vi[k] * ic2) + eps));
        }
        __builtin_nontemporal_store(mi, reinterpret_cast<f32x4*>(m) + i);
        __builtin_nontemporal_store(vi, reinterpret_cast<f32x4*>(v) + i);
        __builtin_nontemporal_store(pi, reinterpret_cast<f32x4*>(p) + i);
        reinterpret_cast<u32x2*>(w16)[i] = u32x2{(uint32_t)f2bf(pi[0]) | ((uint32_t)f2bf(pi[1]) << 16),
                                                 (uint32_t)f2bf(pi[2]) | ((uint32_t)f2bf(pi[3]) << 16)};
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += nth) {
        const float gi = G16 ? bf2f(g16[i]) : g[i];
        float mi = m[i], vi = v[i];
        mi = mi + omb1 * (gi - mi);
        vi = vi + omb2 * (gi * gi - vi);
        m[i] = mi;
        v[i] = vi;
        const float pi = p[i] + (-lr * (mi * ic1) / (sqrtf(vi * ic2) + eps));
        p[i] = pi;
        w16[i] = f2bf(pi);
    }
}
// Adam on one [R, C] weight matrix in TR x TC tiles (TR * TC = 4096, 16 elements per thread): float4 traffic on
// p/g/m/v, 8-B stores to the bf16 copy, and the transposed bf16 copy leaves through an LDS image of the output tile
// like transpose_bf16_kernel.  R % 4 == C % 4 == 0.
template <bool WT, int TC>
__global__ __launch_bounds__(256) void adam_master_bf16_2d_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                  float* __restrict__ m, float* __restrict__ v,
                                                                  bf16_t* __restrict__ w16, bf16_t* __restrict__ wT16,
                                                                  int64_t R, int64_t C, float lr, float b1, float b2,
                                                                  float eps, const double* __restrict__ state,
                                                                  const int* guard) {
    TNN_GUARD_RETURN(guard);
    constexpr int TXN = TC / 4, TR = 4 * (256 / TXN);
    __shared__ __attribute__((aligned(16))) bf16_t tile[WT ? TC : 1][TR + 8];
    const double p1 = state[0], p2 = state[1];
    const float ic1 = (float)(1.0 / (1.0 - p1)), ic2 = (float)(1.0 / (1.0 - p2));
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    const int64_t r0 = (int64_t)blockIdx.y * TR, c0 = (int64_t)blockIdx.x * TC;
    const int tx = threadIdx.x % TXN, ty = threadIdx.x / TXN;
    const int64_t c = c0 + 4 * tx;
    u32x2 row[4];
    auto update = [&](f32x4 gi, f32x4& mi, f32x4& vi, f32x4& pi) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mi[k] = mi[k] + omb1 * (gi[k] - mi[k]);
            vi[k] = vi[k] + omb2 * (gi[k] * gi[k] - vi[k]);
            pi[k] = pi[k] + (-lr * (mi[k] * ic1) / (sqrtf(vi[k] * ic2) + eps));
        }
        return u32x2{(uint32_t)f2bf(pi[0]) | ((uint32_t)f2bf(pi[1]) << 16),
                     (uint32_t)f2bf(pi[2]) | ((uint32_t)f2bf(pi[3]) << 16)};
    };
    if (r0 + TR <= R && c0 + TC <= C) {
        // interior tile: all sixteen 16-B loads in flight before the first use
        const int64_t o0 = (r0 + 4 * ty) * C + c;
        f32x4 gi[4], mi[4], vi[4], pi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // g, m, v and the fp32 master weights are touched once per step: non-temporal on both sides; only the bf16
            // copies are re-read (by the next forward / backward)
            gi[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + o0 + i * C));
            mi[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + o0 + i * C));
            vi[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + o0 + i * C));
            pi[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + o0 + i * C));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            row[i] = update(gi[i], mi[i], vi[i], pi[i]);
            __builtin_nontemporal_store(mi[i], reinterpret_cast<f32x4*>(m + o0 + i * C));
            __builtin_nontemporal_store(vi[i], reinterpret_cast<f32x4*>(v + o0 + i * C));
            __builtin_nontemporal_store(pi[i], reinterpret_cast<f32x4*>(p + o0 + i * C));
            *reinterpret_cast<u32x2*>(w16 + o0 + i * C) = row[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = r0 + 4 * ty + i;
            row[i] = u32x2{0u, 0u};
            if (r < R && c < C) {
                const int64_t o = r * C + c;
                const f32x4 gi = *reinterpret_cast<const f32x4*>(g + o);
                f32x4 mi = *reinterpret_cast<const f32x4*>(m + o);
                f32x4 vi = *reinterpret_cast<const f32x4*>(v + o);
                f32x4 pi = *reinterpret_cast<const f32x4*>(p + o);
                row[i] = update(gi, mi, vi, pi);
                *reinterpret_cast<f32x4*>(m + o) = mi;
                *reinterpret_cast<f32x4*>(v + o) = vi;
                *reinterpret_cast<f32x4*>(p + o) = pi;
                *reinterpret_cast<u32x2*>(w16 + o) = row[i];
            }
        }
    }
    if (!WT) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int w = j >> 1;
        u32x2 col;
        if (j & 1) {
            col.x = (row[0][w] >> 16) | (row[1][w] & 0xffff0000u);
            col.y = (row[2][w] >> 16) | (row[3][w] & 0xffff0000u);
        } else {
            col.x = (row[0][w] & 0xffffu) | (row[1][w] << 16);
            col.y = (row[2][w] & 0xffffu) | (row[3][w] << 16);
        }
        *reinterpret_cast<u32x2*>(&tile[4 * tx + j][4 * ty]) = col;
    }
    __syncthreads();
    constexpr int SEGS = TR / 8;                     // 16-B segments per transposed row
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sidx = (int)threadIdx.x + 256 * i;
        const int cc = sidx / SEGS, seg = sidx % SEGS;
        const int64_t oc = c0 + cc, orow = r0 + seg * 8;
        if (oc < C && orow < R) {
            const u32x4 val = *reinterpret_cast<const u32x4*>(&tile[cc][seg * 8]);
            if (orow + 8 <= R) {
                *reinterpret_cast<u32x4*>(wT16 + oc * R + orow) = val;
            } else {
                *reinterpret_cast<u32x2*>(wT16 + oc * R + orow) = u32x2{val.x, val.y};
            }
        }
    }
}
// Bias of one bf16 Dense layer in ONE launch: db = column sums of dz (bf16 [R, C], core/ops.py:52-54, f32 accumulation like
// colsum_bf16_kernel) and, when p != NULL, Adam on the fp32 master bias + refresh of its bf16 copy (core/optimizer.py:67-79).
// Block = 64 columns x all rows: thread (4 columns, one of 16 row lanes), 8-B loads, the lanes meet in LDS.  Replaces the
// two column-reduction launches + the bias optimizer launch of the bf16 step (3 x ~5-10 us per layer for 32 KB of output).
__device__ __forceinline__ void bias_bf16_block(const int block, const bf16_t* __restrict__ dz, int64_t R, int64_t C,
                                                float* __restrict__ db, float* __restrict__ p, float* __restrict__ m,
                                                float* __restrict__ v, bf16_t* __restrict__ w16, float lr, float b1,
                                                float b2, float eps, const double* __restrict__ state,
                                                const int* guard) {
    __shared__ float part[16][64 + 4];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int64_t c0 = (int64_t)block * 64 + 4 * cq;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 + 3 < C && C % 4 == 0 && (reinterpret_cast<uintptr_t>(dz) & 7) == 0) {
        constexpr int UN = 8;
        for (int64_t r0 = rl; r0 < R; r0 += 16 * UN) {
            u32x2 q[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t r = r0 + 16 * u;
                q[u] = r < R ? *reinterpret_cast<const u32x2*>(dz + r * C + c0) : u32x2{0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                a[0] += __uint_as_float(q[u].x << 16); a[1] += __uint_as_float(q[u].x & 0xffff0000u);
                a[2] += __uint_as_float(q[u].y << 16); a[3] += __uint_as_float(q[u].y & 0xffff0000u);
            }
        }
    } else {
        for (int e = 0; e < 4; ++e)
            if (c0 + e < C)
                for (int64_t r = rl; r < R; r += 16) a[e] += bf2f(dz[r * C + c0 + e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[rl][4 * cq + e] = a[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int64_t c = (int64_t)block * 64 + threadIdx.x;
        if (c < C) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += part[q][threadIdx.x];
            db[c] = s;
            if (p != nullptr && (guard == nullptr || __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)) {
                const float ic1 = (float)(1.0 / (1.0 - state[0])), ic2 = (float)(1.0 / (1.0 - state[1]));
                float mi = m[c], vi = v[c];
                mi = mi + (1.f - b1) * (s - mi);
                vi = vi + (1.f - b2) * (s * s - vi);
                m[c] = mi;
                v[c] = vi;
                const float pi = p[c] + (-lr * (mi * ic1) / (sqrtf(vi * ic2) + eps));
                p[c] = pi;
                if (w16 != nullptr) w16[c] = f2bf(pi);
            }
        }
    }
}

__global__ __launch_bounds__(256) void bias_bf16_kernel(const bf16_t* __restrict__ dz, int64_t R, int64_t C,
                                                        float* __restrict__ db, float* __restrict__ p, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ w16, float lr, float b1,
                                                        float b2, float eps, const double* __restrict__ state,
                                                        const int* guard) {
    bias_bf16_block((int)blockIdx.x, dz, R, C, db, p, m, v, w16, lr, b1, b2, eps, state, guard);
}
// the biases of SEVERAL layers in one launch (tnn_bias_bf16_adam_multi: the last launch of the single-GPU bf16 step): block b
// belongs to the layer whose block range holds it and does exactly what bias_bf16_kernel's block does — the same bits
struct BiasMulti {
    static constexpr int MAXL = 16;
    int n;
    int first_block[MAXL + 1];
    const bf16_t* dz[MAXL];
    int64_t C[MAXL];
    float *db[MAXL], *p[MAXL], *m[MAXL], *v[MAXL];
    bf16_t* w16[MAXL];
};
__global__ __launch_bounds__(256) void bias_bf16_multi_kernel(BiasMulti a, int64_t R, float lr, float b1, float b2, float eps,
                                                              const double* __restrict__ state, const int* guard) {
    int l = 0;
    while (l + 1 < a.n && (int)blockIdx.x >= a.first_block[l + 1]) ++l;      // block-uniform
    bias_bf16_block((int)blockIdx.x - a.first_block[l], a.dz[l], R, a.C[l], a.db[l], a.p[l], a.m[l], a.v[l], a.w16[l], lr, b1, b2,
                    eps, state, guard);
}

__global__ void adam_advance16_kernel(double* __restrict__ state, double b1, double b2, const int* guard) {
    TNN_GUARD_RETURN(guard);
    state[0] *= b1;
    state[1] *= b2;
}

}  // namespace

extern "C" {

// the skinny-product test of tnn_gemm_bf16_nt: does (M, N, K) -> bf16 take the split-K kernel on this chip?
static bool sk_shape(int64_t M, int64_t N, int64_t K, int64_t* t256_out) {
    constexpr int S = 2, SK_BN = 128, SK_NSB = 4;
    const int64_t t256 = ((M + 255) / 256) * ((N + SK_BN - 1) / SK_BN), nk = K / 64;
    const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int cus = tnn::num_cus();
    if (t256_out) *t256_out = t256;
    return nk % S == 0 && nk / S > SK_NSB && t256 * S <= cus && t256 * S * 2 >= cus && tiles <= 2 * (int64_t)cus;
}

static int gemm_bf16_nt_impl(const char* fn, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                             void* C, int64_t ldc, int c_dtype, const void* bias_f32, int act, int relu_sign,
                             const void* mask_y, int64_t ldy, void* CT, int64_t ldct);

int tnn_gemm_bf16_nt(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                     void* C, int64_t ldc, int c_dtype, const void* bias_f32, int act, int relu_sign,
                     const void* mask_y, int64_t ldy) {
    return gemm_bf16_nt_impl("tnn_gemm_bf16_nt", M, N, K, A, lda, B, ldb, C, ldc, c_dtype, bias_f32, act, relu_sign, mask_y, ldy,
                             nullptr, 0);
}

int tnn_gemm_bf16_nt_t(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                       void* C, int64_t ldc, const void* bias_f32, int act, int relu_sign, const void* mask_y, int64_t ldy,
                       void* C_t, int64_t ldct) {
    TNN_REQUIRE(C_t != nullptr && ldct >= M, "tnn_gemm_bf16_nt_t: C_t is required, with ldct >= M");
    return gemm_bf16_nt_impl("tnn_gemm_bf16_nt_t", M, N, K, A, lda, B, ldb, C, ldc, TNN_BF16, bias_f32, act, relu_sign, mask_y, ldy,
                             C_t, ldct);
}

int tnn_gemm_bf16_reserve(int64_t M, int64_t N, int64_t K) {
    TNN_NEED_INIT();
    int64_t t256 = 0;
    if (!sk_shape(M, N, K, &t256)) return 0;
    SkWorkspace w;
    TNN_REQUIRE(sk_workspace(tnn::stream(), (int)t256, 2, (size_t)256 * 128 * 4, &w),
                "tnn_gemm_bf16_reserve: could not allocate the split-K hand-off memory (inside a capture, or out of memory)");
    return 0;
}

static int gemm_bf16_nt_impl(const char* fn, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                             void* C, int64_t ldc, int c_dtype, const void* bias_f32, int act, int relu_sign,
                             const void* mask_y, int64_t ldy, void* CT, int64_t ldct) {
    (void)fn;
    TNN_NEED_INIT();
    TNN_REQUIRE(M > 0 && N > 0 && K > 0, "tnn_gemm_bf16_nt: empty problem");
    TNN_REQUIRE(K % BK == 0, "tnn_gemm_bf16_nt: K (%lld) must be a multiple of %d", (long long)K, BK);
    TNN_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K, "tnn_gemm_bf16_nt: lda/ldb must be >= K and multiples of 8");
    TNN_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0,
                "tnn_gemm_bf16_nt: operands must be 16-byte aligned");
    TNN_REQUIRE(M * lda * 2 < (int64_t(1) << 32) && N * ldb * 2 < (int64_t(1) << 32),
                "tnn_gemm_bf16_nt: operands of 4 GiB or more are not supported (32-bit buffer offsets)");
    TNN_REQUIRE(c_dtype == TNN_F32 || c_dtype == TNN_BF16, "tnn_gemm_bf16_nt: output dtype %d", c_dtype);
    TNN_REQUIRE(ldc >= N, "tnn_gemm_bf16_nt: ldc too small");
    TNN_REQUIRE(!(bias_f32 || act) || !mask_y, "tnn_gemm_bf16_nt: bias/activation and mask epilogues are exclusive");
    BfArgs g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.c_bf16 = c_dtype == TNN_BF16;
    g.epi = mask_y ? BEPI_MASK : (bias_f32 || act) ? BEPI_BIAS_ACT : BEPI_PLAIN;
    g.bias = (const float*)bias_f32; g.act = act; g.relu_sign = relu_sign;
    g.Y = (const bf16_t*)mask_y; g.ldy = ldy;
    g.tiles_m = (int)((M + BM - 1) / BM);
    g.tiles_n = (int)((N + BN - 1) / BN);
    // Kernel choice — all LDS-DMA now.  Grids with at least two tiles per CU (the dW products, large squares): 8 waves and
    // a 2-stage ring (64 KB of LDS, TWO workgroups per CU: 597 vs 503 TFLOP/s on dW, 978 vs 934 on 4096^3).  One tile per CU
    // (the M = 512 forward / dX shapes): 8 waves and a 4-stage ring.  With ONE weight matrix re-used by every call — what a
    // micro-benchmark does — the register-staged 4-wave kernel looked equal (810 vs 806 TFLOP/s: 134 MB of bf16 weights
    // stay in the 256 MB memory-side cache); in the training step every layer streams its own W / W^T from HBM and the
    // deeper ring wins (whole config-E step: 216 k -> 237 k samples/s; bench.py's per-GEMM table rotates operands for this
    // reason).  Variants: reg | dma8s | dma4s | dma8 | dma4 (8 / 4 waves; s = 2 stages).
    const unsigned tiles = (unsigned)(g.tiles_m * g.tiles_n);
    // Skinny products (at most about one 128 x 128 tile per CU: config E's M = 512 forward / dX shapes, 87 us at 512 x 8192 x
    // 8192): 256 x 128 tiles with the K range split over two workgroups, 70 us (tnn_gemm_bf16_sk.h; 256 x 256 x 4 slices loses
    // more to its 48 MB slab exchange than its K loop gains).  One workgroup per CU, so the grid must fit the chip once.
    {
        constexpr int S = 2, SK_BN = 128, SK_NSB = 4;
        int64_t t256 = 0;
        if (g.c_bf16 && sk_shape(M, N, K, &t256)) {
            SkWorkspace w;
            // (the hand-off memory is allocated on first use OUTSIDE a capture; tnn_gemm_bf16_reserve — called by the trainer
            // when it is created — does that up front, so an eager step and a captured one run the same kernel)
            if (sk_workspace(tnn::stream(), (int)t256, S, (size_t)256 * SK_BN * 4, &w)) {
                g.tiles_m = (int)((M + 255) / 256);
                g.tiles_n = (int)((N + SK_BN - 1) / SK_BN);
                g.splitk = S;
                g.sk_ws = w.slabs;
                g.sk_cnt = w.counters;
                g.CT = (bf16_t*)CT; g.ldct = ldct;
                g.fault = tnn::fault_word();
                hipLaunchKernelGGL((sk::gemm_bf16_sk_kernel<SK_BN, 4, 2, 3, SK_NSB>), dim3((unsigned)(t256 * S)), 512, 0,
                                   tnn::stream(), g);
                TNN_LAUNCH_OK();
                return 0;
            }
        }
    }
    const char* which = tiles >= 2u * (unsigned)tnn::num_cus() ? "dma8s" : "dma8";
    const bool dma = which[0] == 'd' && which[1] == 'm' && which[2] == 'a';
    const bool swap = g.c_bf16 != 0;          // bf16 outputs: 8-B stores per lane; fp32 outputs: whole 128-B row segments
#define TNN_DMA_LAUNCH(NW_, NS_)                                                                                         \
    do {                                                                                                                 \
        if (swap) hipLaunchKernelGGL((gemm_bf16_dma_kernel<NW_, NS_, true>), dim3(tiles), NW_ * 64, 0, tnn::stream(), g);  \
        else hipLaunchKernelGGL((gemm_bf16_dma_kernel<NW_, NS_, false>), dim3(tiles), NW_ * 64, 0, tnn::stream(), g);      \
    } while (0)
    if (dma && which[3] == '8' && which[4] == 's') TNN_DMA_LAUNCH(8, 2);
    else if (dma && which[3] == '4' && which[4] == 's') TNN_DMA_LAUNCH(4, 2);
    else if (dma && which[3] == '8') TNN_DMA_LAUNCH(8, 4);
    else if (dma) TNN_DMA_LAUNCH(4, 4);
#undef TNN_DMA_LAUNCH
    else
        hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3(tiles), NT, 0, tnn::stream(), g);
    TNN_LAUNCH_OK();
    // the transposed second output on a shape the split-K kernel does not take: a launch of its own (needs C dense)
    if (CT != nullptr) {
        TNN_REQUIRE(ldc == N && ldct == M, "%s: the transposed output of this shape needs dense C and C_t", fn);
        return tnn_transpose_bf16(C, CT, M, N);
    }
    return 0;
}

int tnn_gemm_bf16_nt_adam(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                          void* g_out_f32, void* p_master, void* m, void* v, void* w_bf16, void* wT_bf16, double lr,
                          double b1, double b2, double eps, const void* pows_f64) {
    TNN_NEED_INIT();
    TNN_REQUIRE(M > 0 && N > 0 && K > 0, "tnn_gemm_bf16_nt_adam: empty problem");
    TNN_REQUIRE(K % BK == 0, "tnn_gemm_bf16_nt_adam: K (%lld) must be a multiple of %d", (long long)K, BK);
    TNN_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K, "tnn_gemm_bf16_nt_adam: lda/ldb must be >= K and multiples of 8");
    TNN_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0,
                "tnn_gemm_bf16_nt_adam: operands must be 16-byte aligned");
    TNN_REQUIRE(M * lda * 2 < (int64_t(1) << 32) && N * ldb * 2 < (int64_t(1) << 32),
                "tnn_gemm_bf16_nt_adam: operands of 4 GiB or more are not supported (32-bit buffer offsets)");
    TNN_REQUIRE(p_master && m && v && pows_f64, "tnn_gemm_bf16_nt_adam: p, m, v and pows are required");
    TNN_REQUIRE(wT_bf16 == nullptr || (M % 4 == 0 && (reinterpret_cast<uintptr_t>(wT_bf16) & 7) == 0),
                "tnn_gemm_bf16_nt_adam: the transposed copy needs M %% 4 == 0 and 8-byte alignment");
    BfArgs g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = g_out_f32;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N;
    g.c_bf16 = 0;
    g.epi = BEPI_PLAIN;
    g.tiles_m = (int)((M + BM - 1) / BM);
    g.tiles_n = (int)((N + BN - 1) / BN);
    g.ap = (float*)p_master; g.am = (float*)m; g.av = (float*)v;
    g.aw16 = (bf16_t*)w_bf16; g.awT16 = (bf16_t*)wT_bf16; g.ldt = M;
    g.lr = (float)lr; g.b1 = (float)b1; g.b2 = (float)b2; g.eps = (float)eps;
    g.pows = (const double*)pows_f64;
    g.guard = tnn::update_guard();
    const unsigned tiles = (unsigned)(g.tiles_m * g.tiles_n);
    hipLaunchKernelGGL((gemm_bf16_dma_kernel<8, 2, false, true>), dim3(tiles), 8 * 64, 0, tnn::stream(), g);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_bias_bf16_adam_multi(int n_layers, const void* const* dz, int64_t rows, const int64_t* cols, void* const* db_f32,
                             void* const* p_master, void* const* m, void* const* v, void* const* w_bf16, double lr, double b1,
                             double b2, double eps, const void* pows_f64) {
    TNN_NEED_INIT();
    TNN_REQUIRE(n_layers >= 1 && n_layers <= BiasMulti::MAXL && dz && cols && db_f32 && rows > 0,
                "tnn_bias_bf16_adam_multi: 1 .. %d layers, dz / cols / db required", BiasMulti::MAXL);
    TNN_REQUIRE((p_master != nullptr) == (m != nullptr) && (p_master != nullptr) == (v != nullptr) &&
                    (p_master == nullptr || pows_f64 != nullptr),
                "tnn_bias_bf16_adam_multi: p / m / v / pows go together");
    BiasMulti a = {};
    a.n = n_layers;
    int blocks = 0;
    for (int l = 0; l < n_layers; ++l) {
        TNN_REQUIRE(dz[l] && db_f32[l] && cols[l] > 0, "tnn_bias_bf16_adam_multi: layer %d: dz, db and cols are required", l);
        a.dz[l] = (const bf16_t*)dz[l]; a.C[l] = cols[l]; a.db[l] = (float*)db_f32[l];
        a.p[l] = p_master ? (float*)p_master[l] : nullptr;
        a.m[l] = m ? (float*)m[l] : nullptr;
        a.v[l] = v ? (float*)v[l] : nullptr;
        a.w16[l] = w_bf16 ? (bf16_t*)w_bf16[l] : nullptr;
        a.first_block[l] = blocks;
        blocks += (int)((cols[l] + 63) / 64);
    }
    a.first_block[n_layers] = blocks;
    hipLaunchKernelGGL(bias_bf16_multi_kernel, dim3((unsigned)blocks), 256, 0, tnn::stream(), a, rows, (float)lr, (float)b1, (float)b2,
                       (float)eps, (const double*)pows_f64, tnn::update_guard());
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_mse_bf16_prep(const void* pred, const void* y, int64_t rows, int64_t cols, int64_t m_global, void* loss_out_f32,
                      void* loss_out2_f32, void* dpred, void* dpred_t, const void* x, int64_t x_cols, void* x_t,
                      void* partials_f64, void* ticket_u32, void* adam_pows_f64, double b1, double b2) {
    TNN_NEED_INIT();
    TNN_REQUIRE(pred && y && rows > 0 && cols > 0 && m_global > 0, "tnn_mse_bf16_prep: empty batch");
    TNN_REQUIRE(rows % 64 == 0 && cols % 64 == 0 && (x == nullptr || x_cols % 64 == 0),
                "tnn_mse_bf16_prep: rows, cols and x_cols must be multiples of 64");
    TNN_REQUIRE((x == nullptr) == (x_t == nullptr), "tnn_mse_bf16_prep: x and x_t go together");
    TNN_REQUIRE(partials_f64 && ticket_u32, "tnn_mse_bf16_prep: the partial-sum workspace (rows / 64 * cols / 64 doubles) and the 64 zeroed ticket words are required");
    TNN_REQUIRE(loss_out_f32 != nullptr || loss_out2_f32 == nullptr, "tnn_mse_bf16_prep: loss_out2 needs loss_out");
    auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    TNN_REQUIRE(al(pred, 8) && al(y, 8) && al(dpred, 8) && al(dpred_t, 16) && al(x, 8) && al(x_t, 16), "tnn_mse_bf16_prep: misaligned operand");
    const int64_t n_loss = (rows / 64) * (cols / 64), n_x = x ? (rows / 64) * (x_cols / 64) : 0;
    TNN_REQUIRE(n_loss + n_x < (int64_t(1) << 31), "tnn_mse_bf16_prep: too many tiles");
    hipLaunchKernelGGL(mse_prep_bf16_kernel, dim3((unsigned)(n_loss + n_x)), 256, 0, tnn::stream(), (const bf16_t*)pred,
                       (const bf16_t*)y, rows, cols, 1.0 / (double)m_global, (double*)partials_f64, (unsigned*)ticket_u32,
                       (float*)loss_out_f32, (float*)loss_out2_f32, (bf16_t*)dpred, (bf16_t*)dpred_t, (const bf16_t*)x, x_cols,
                       (bf16_t*)x_t, (double*)adam_pows_f64, b1, b2, tnn::update_guard());
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_adam_tick(void* pows_f64, double b1, double b2) {
    TNN_NEED_INIT();
    TNN_REQUIRE(pows_f64 != nullptr, "tnn_adam_tick: pows state is NULL");
    hipLaunchKernelGGL(adam_advance16_kernel, 1, 1, 0, tnn::stream(), (double*)pows_f64, b1, b2, tnn::update_guard());
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_transpose_bf16(const void* in, void* out, int64_t rows, int64_t cols) {
    TNN_NEED_INIT();
    if (rows * cols <= 0) return 0;
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    TNN_REQUIRE(grid.y <= 65535, "tnn_transpose_bf16: too many rows");
    const bool fast = rows % 4 == 0 && cols % 4 == 0 &&
                      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (fast)
        hipLaunchKernelGGL(transpose_bf16_kernel, grid, 256, 0, tnn::stream(), (const bf16_t*)in, (bf16_t*)out, rows, cols);
    else
        hipLaunchKernelGGL(transpose_bf16_slow_kernel, grid, 256, 0, tnn::stream(), (const bf16_t*)in, (bf16_t*)out, rows, cols);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_transpose2_bf16(const void* in1, void* out1, int64_t rows1, int64_t cols1, const void* in2, void* out2, int64_t rows2,
                        int64_t cols2) {
    TNN_NEED_INIT();
    auto fast = [](const void* a, const void* b, int64_t r, int64_t c) {
        return r > 0 && c > 0 && r % 4 == 0 && c % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    };
    if (!fast(in1, out1, rows1, cols1) || !fast(in2, out2, rows2, cols2)) {          // odd shapes: the two launches
        if (int rc = tnn_transpose_bf16(in1, out1, rows1, cols1)) return rc;
        return tnn_transpose_bf16(in2, out2, rows2, cols2);
    }
    const int64_t n1 = ((rows1 + 63) / 64) * ((cols1 + 63) / 64), n2 = ((rows2 + 63) / 64) * ((cols2 + 63) / 64);
    TNN_REQUIRE(n1 + n2 < (int64_t(1) << 31), "tnn_transpose2_bf16: too many tiles");
    hipLaunchKernelGGL(transpose2_bf16_kernel, dim3((unsigned)(n1 + n2)), 256, 0, tnn::stream(), (const bf16_t*)in1, (bf16_t*)out1, rows1,
                       cols1, (const bf16_t*)in2, (bf16_t*)out2, rows2, cols2, (int)n1);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_cast_bf16(const void* in, void* out, int64_t n, int to_bf16) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    unsigned grid = tnn::stream_grid(n, 256);
    if (to_bf16)
        hipLaunchKernelGGL(cast_f32_bf16_kernel, grid, 256, 0, tnn::stream(), (const float*)in, (bf16_t*)out, n);
    else
        hipLaunchKernelGGL(cast_bf16_f32_kernel, grid, 256, 0, tnn::stream(), (const bf16_t*)in, (float*)out, n);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_colsum_bf16(const void* in, void* out_f32, int64_t rows, int64_t cols) {
    // one launch, the summation order of tnn_bias_bf16_adam (so a step that fuses the bias optimizer in and one that does
    // not produce the same bits)
    return tnn_bias_bf16_adam(in, rows, cols, out_f32, nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, 0.0, nullptr);
}

int tnn_mse_bf16(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out_f32, void* dpred) {
    return tnn_mse_bf16_tick(pred, y, n, m_global, loss_out_f32, nullptr, dpred, nullptr, 1.0, 1.0);
}

int tnn_bias_bf16_adam(const void* dz, int64_t rows, int64_t cols, void* db_f32, void* p_master, void* m, void* v, void* w_bf16,
                       double lr, double b1, double b2, double eps, const void* pows_f64) {
    TNN_NEED_INIT();
    if (cols <= 0) return 0;
    TNN_REQUIRE(dz && db_f32 && rows > 0, "tnn_bias_bf16_adam: dz and db are required");
    TNN_REQUIRE((p_master == nullptr) == (m == nullptr) && (p_master == nullptr) == (v == nullptr) &&
                    (p_master == nullptr || pows_f64 != nullptr),
                "tnn_bias_bf16_adam: p / m / v / pows go together");
    hipLaunchKernelGGL(bias_bf16_kernel, dim3((unsigned)((cols + 63) / 64)), 256, 0, tnn::stream(), (const bf16_t*)dz, rows, cols,
                       (float*)db_f32, (float*)p_master, (float*)m, (float*)v, (bf16_t*)w_bf16, (float)lr, (float)b1, (float)b2,
                       (float)eps, (const double*)pows_f64, tnn::update_guard());
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_mse_bf16_tick(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out_f32, void* loss_out2_f32,
                      void* dpred, void* adam_pows_f64, double b1, double b2) {
    TNN_NEED_INIT();
    TNN_REQUIRE(n > 0 && m_global > 0, "tnn_mse_bf16: empty batch");
    TNN_REQUIRE(loss_out_f32 != nullptr || loss_out2_f32 == nullptr, "tnn_mse_bf16_tick: loss_out2 needs loss_out");
    int64_t nb = tnn::stream_grid(n, 256);
    if (nb > 1024) nb = 1024;
    void* ws = nullptr;
    if (tnn_malloc((size_t)nb * sizeof(double), &ws)) return 1;
    hipLaunchKernelGGL(mse_bf16_kernel, (unsigned)nb, 256, 0, tnn::stream(), (const bf16_t*)pred, (const bf16_t*)y, n,
                       1.0 / (double)m_global, (double*)ws, (bf16_t*)dpred, (double*)adam_pows_f64, b1, b2, tnn::update_guard());
    if (loss_out_f32)
        hipLaunchKernelGGL(sum_partials_f32_kernel, 1, 64, 0, tnn::stream(), (const double*)ws, (int)nb,
                           (float*)loss_out_f32, (float*)loss_out2_f32);
    tnn_free(ws);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_adam_master_bf16(void* p_master, const void* g, void* m, void* v, void* w_bf16, int64_t n, double lr,
                         double b1, double b2, double eps, void* pows_f64) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    TNN_REQUIRE(pows_f64 != nullptr, "tnn_adam_master_bf16: pows state is NULL");
    hipStream_t s = tnn::stream();
    hipLaunchKernelGGL(adam_advance16_kernel, 1, 1, 0, s, (double*)pows_f64, b1, b2, tnn::update_guard());
    hipLaunchKernelGGL(adam_master_bf16_kernel<false>, tnn::stream_grid((n + 3) / 4, 256), 256, 0, s, (float*)p_master,
                       (const float*)g, (float*)m, (float*)v, (bf16_t*)w_bf16, n, (float)lr, (float)b1, (float)b2,
                       (float)eps, (const double*)pows_f64, tnn::update_guard());
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_adam_master_g16(void* p_master, const void* g_bf16, void* m, void* v, void* w_bf16, int64_t n, double lr, double b1,
                        double b2, double eps, const void* pows_f64) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    TNN_REQUIRE(p_master && g_bf16 && m && v && w_bf16 && pows_f64, "tnn_adam_master_g16: NULL argument");
    hipLaunchKernelGGL(adam_master_bf16_kernel<true>, tnn::stream_grid((n + 3) / 4, 256), 256, 0, tnn::stream(),
                       (float*)p_master, g_bf16, (float*)m, (float*)v, (bf16_t*)w_bf16, n, (float)lr, (float)b1, (float)b2,
                       (float)eps, (const double*)pows_f64, tnn::update_guard());
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_adam_master_bf16_2d(void* p_master, const void* g, void* m, void* v, void* w_bf16, void* wT_bf16,
                            int64_t rows, int64_t cols, double lr, double b1, double b2, double eps, void* pows_f64,
                            int advance) {
    TNN_NEED_INIT();
    TNN_REQUIRE(pows_f64 != nullptr, "tnn_adam_master_bf16_2d: pows state is NULL");
    hipStream_t s = tnn::stream();
    if (advance) hipLaunchKernelGGL(adam_advance16_kernel, 1, 1, 0, s, (double*)pows_f64, b1, b2, tnn::update_guard());
    if (rows <= 0 || cols <= 0) {
        TNN_LAUNCH_OK();
        return 0;
    }
    const uintptr_t align = reinterpret_cast<uintptr_t>(p_master) | reinterpret_cast<uintptr_t>(g) |
                            reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                            reinterpret_cast<uintptr_t>(wT_bf16) | (reinterpret_cast<uintptr_t>(w_bf16) << 1);
    if (rows % 4 == 0 && cols % 4 == 0 && (align & 15) == 0) {
        constexpr int TC = 64, TR = 64;      // measured on 8192x8192: 64x64 3.15 ms/step, 32x128 (wider rows) 3.31
        dim3 grid((unsigned)((cols + TC - 1) / TC), (unsigned)((rows + TR - 1) / TR));
        if (wT_bf16)
            hipLaunchKernelGGL((adam_master_bf16_2d_kernel<true, TC>), grid, 256, 0, s, (float*)p_master, (const float*)g,
                               (float*)m, (float*)v, (bf16_t*)w_bf16, (bf16_t*)wT_bf16, rows, cols, (float)lr,
                               (float)b1, (float)b2, (float)eps, (const double*)pows_f64, tnn::update_guard());
        else
            hipLaunchKernelGGL((adam_master_bf16_2d_kernel<false, TC>), grid, 256, 0, s, (float*)p_master, (const float*)g,
                               (float*)m, (float*)v, (bf16_t*)w_bf16, (bf16_t*)nullptr, rows, cols, (float)lr,
                               (float)b1, (float)b2, (float)eps, (const double*)pows_f64, tnn::update_guard());
        TNN_LAUNCH_OK();
        return 0;
    }
    const int64_t n = rows * cols;
    hipLaunchKernelGGL(adam_master_bf16_kernel<false>, tnn::stream_grid(n, 256), 256, 0, s, (float*)p_master, (const float*)g,
                       (float*)m, (float*)v, (bf16_t*)w_bf16, n, (float)lr, (float)b1, (float)b2, (float)eps,
                       (const double*)pows_f64, tnn::update_guard());
    TNN_LAUNCH_OK();
    if (wT_bf16) return tnn_transpose_bf16(w_bf16, wT_bf16, rows, cols);
    return 0;
}

}  // extern "C"
