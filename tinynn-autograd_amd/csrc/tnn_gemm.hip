// K1: GEMM for the three contractions of ops.dot_ (core/ops.py:151 NN, :157 NT, :160 TN) with the
// fused epilogues of K8.  fp32 path = v_mfma_f32_32x32x2_f32 (exact f32, 256 FLOP/clk/CU, 157 TF
// chip peak), LDS double-buffered, one barrier per K-tile, register-staged prefetch of the next
// tile, XCD-aware tile order.  No operand is ever transposed in memory: each operand is staged in
// the layout it already has and only the LDS read pattern differs.
//
// Operand layouts ("KC" = K is the contiguous axis of the stored matrix):
//     NN: A[M,K] KC      B[K,N] N-contiguous
//     NT: A[M,K] KC      B[N,K] KC
//     TN: A[K,M] M-contig B[K,N] N-contiguous
// LDS images:  KC operand      -> [rows][BK+4]   fragment = one ds_read_b128 per 8-deep k-chunk
//                                  (row stride 144 B: conflict-free for the 16-lane b128 groups)
//              MN-contig operand -> [BK][rows]    fragment = 4 ds_read_b32 (32 consecutive dwords)
// k-assignment inside an 8-deep chunk: lanes 0-31 hold k = 0..3, lanes 32-63 hold k = 4..7; MFMA j
// contracts {j, 4+j}.  A and B use the same assignment, so the sum is complete; only the fp32
// summation order differs from a sequential loop (tolerance, not bit-exactness, is the f32 bar).
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "tnn_internal.h"
#include "tnn_p2p.h"
#include "tnn_head_stats.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_AXPBY = 0, EPI_BIAS_ACT = 1, EPI_MASK = 2, EPI_ADAM = 3 };
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int64_t M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int epi;
    const float* bias;
    int act, relu_sign;
    const float* Y;
    int64_t ldy;
    int vecA, vecB;        // 16-B vector loads legal for this operand
    int64_t k_per_split;   // multiple of BK
    int splits;
    float* ws;             // [splits, M, N] partial sums when splits > 1
    int tiles_m, tiles_n;
    // small/latency kernel: XCD-aware tile order.  Workgroup b runs on XCD b % 8 and each XCD has a private L2; with
    // xg_m x xg_n == 8 the tile grid is cut into 8 rectangles of (tiles_m / xg_m) x (tiles_n / xg_n) tiles, one per XCD,
    // so an L2 fills only the operand rows / columns of its rectangle.  0 = plain order.
    int xg_m, xg_n;
    // small/latency kernel only (tnn_dense_fwd_head_partials): the NEXT (classifier) layer's weights head_w [N, head_c]
    // and where this tile's share of that layer's logits goes, head_z [tiles_n][M][head_c]
    const float* head_w;
    float* head_z;
    int head_c;
    // small/latency kernel, FAST form only (tnn_dense_fwd_head_partials_stats with exchange = 2): a launch sequence that ONE
    // thread of the launch advances — the tag of the deferred statistics exchange in the head launch behind it (tnn_p2p.h: XchgCtx)
    unsigned int* bump;
    // EPI_ADAM (tiled kernel, tnn_gemm_tn_adam): the product is a weight gradient that Adam consumes in the epilogue;
    // C (may be NULL) receives the gradient itself
    float *ad_p, *ad_m, *ad_v;
    float ad_lr, ad_b1, ad_b2, ad_eps;
    const double* ad_pows;
    const int* ad_guard;
    int staged_c;          // tiled kernel: interior tiles store through an LDS image of the tile (row-major 16-B stores)
    int group_m;           // tiled kernel: M-tiles per raster group (8; 1 = an M panel per XCD range, tiles_m = N-major sweep)
    // tiled TN kernel with EPI_ADAM (tnn_gemm_tn_adam_bias): the workgroups of tile row 0 also produce the column sums of B
    // (= the bias gradient, core/ops.py:52-54) in their epilogue -> cs_db [N], and apply Adam to the bias block
    // cs_p / cs_m / cs_v [N] when those are given
    float *cs_db, *cs_p, *cs_m, *cs_v;
#ifdef TNN_GEMM_TRACE
    unsigned long long* trace;   // debug build only: [grid][8] timeline words (nullptr = off)
#endif
};

__device__ __forceinline__ float apply_epilogue(const GemmArgs& g, float acc, int64_t row,
                                                int64_t col) {
    if (g.epi == EPI_AXPBY) {
        float r = g.alpha * acc;
        if (g.beta != 0.f) r += g.beta * g.C[row * g.ldc + col];
        return r;
    } else if (g.epi == EPI_BIAS_ACT) {
        float v = acc + (g.bias ? g.bias[col] : 0.f);
        if (g.act == TNN_ACT_RELU) {
            if (g.relu_sign) v = v < 0.f ? -0.0f : fabsf(v);   // mask x>=0 kept in the sign bit
            else v = v < 0.f ? 0.f : v;
        }
        return v;
    } else {
        float y = g.Y[row * g.ldy + col];
        return (__float_as_uint(y) >> 31) ? 0.f : acc;
    }
}

// bijective XCD remap (blocks b, b+8, b+16 ... share an XCD and therefore an L2): give every XCD a
// contiguous range of tile ids so neighbouring tiles (same B panel, adjacent A panels) hit in L2.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int nx = 8;
    if (nb < 2 * nx) return b;
    int q = nb / nx, r = nb % nx;
    int xcd = b % nx, local = b / nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// VEC = both operands allow 16-B loads (aligned base, leading dimension and contiguous extent multiples
// of 4): the main loop is then branch-free — per-thread source pointers, row-validity flags and LDS
// offsets are computed once before the loop, out-of-range rows read a clamped (valid) address and are
// zeroed with a select, and only the last, partial K-tile goes through the guarded element-wise loader.
// VEC = false is the fully guarded element-wise variant for odd shapes / unaligned views.
template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, bool VEC>
__global__ __launch_bounds__(WM* WN * 64) void gemm_f32_mfma_kernel(GemmArgs g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int SA = AKC ? BK + 4 : BM;          // LDS row stride (floats)
    constexpr int SB = BKC ? BK + 4 : BN;
    constexpr int A_ELEMS = AKC ? BM * SA : BK * SA;
    constexpr int B_ELEMS = BKC ? BN * SB : BK * SB;
    constexpr int A_F4 = BM * BK / 4 / NT;         // float4 per thread per tile
    constexpr int B_F4 = BN * BK / 4 / NT;
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for the block");
    static_assert(BK % 8 == 0, "BK must be a multiple of 8");

    __shared__ __attribute__((aligned(16))) float lds[2 * (A_ELEMS + B_ELEMS)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    // Debug build only (make trace -> libtnn_hip_trace.so, tools/gemm_block_timeline.py): thread 0 of every
    // workgroup leaves four 100 MHz timestamps (entry, K loop start, K loop end, stores acknowledged) and its
    // HW_ID / XCC_ID in a host-supplied buffer, 8 words per workgroup.
#ifdef TNN_GEMM_TRACE
#define TNN_TRACE(i) do { if (tid == 0 && g.trace) g.trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
    if (tid == 0 && g.trace) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g.trace[(size_t)blockIdx.x * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
    }
#else
#define TNN_TRACE(i)
#endif
    TNN_TRACE(0);
    // tile order: each XCD (private 4 MB L2) gets a contiguous range of ids (xcd_remap); inside the range the
    // ids sweep N for a GROUP of 8 M-tiles at a time, so the group's A panels stay L2-resident while the B
    // panels stream through once (measured before this: 469 MB fetched by the 4096x4096x512 TN GEMM for
    // 16 MB of inputs, every A panel was evicted between its reuses)
    const int nb = g.tiles_m * g.tiles_n;
    const int t = xcd_remap((int)blockIdx.x, nb);
    const int GROUP_M = g.group_m;                     // 8 (TNN_GEMM_GROUP_M: the raster sweep of tools/gemm_sweep.py)
    const int per_group = GROUP_M * g.tiles_n;
    const int first_m = (t / per_group) * GROUP_M;
    const int gsz = min(g.tiles_m - first_m, GROUP_M);
    const int64_t m0 = (int64_t)(first_m + (t % per_group) % gsz) * BM;
    const int64_t n0 = (int64_t)((t % per_group) / gsz) * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
    const int64_t kend = min(g.K, kbeg + g.k_per_split);
    const int nk = (int)((kend - kbeg + BK - 1) / BK);
    const int nk_full = (int)((kend - kbeg) / BK);      // tiles that need no K guard

    // ---- per-thread staging geometry, fixed for the whole K loop
    // element (row, c4) of the tile: KC operand -> row = m/n index, c4 = k/4;  MN-contig -> row = k, c4 = m/4
    // fast loader addressing = wave-uniform base (advances by a scalar add per K-tile) + a per-thread 32-bit byte
    // offset that never changes (the host takes this kernel's VEC variant only for operands below 4 GiB)
    uint32_t a_off[A_F4], b_off[B_F4];
    int a_dst[A_F4], b_dst[B_F4];
    int a_row[A_F4], a_c4[A_F4], b_row[B_F4], b_c4[B_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int f = tid + i * NT;
        a_row[i] = AKC ? f / (BK / 4) : f / (BM / 4);
        a_c4[i] = AKC ? f % (BK / 4) : f % (BM / 4);
        a_dst[i] = a_row[i] * SA + a_c4[i] * 4;
        if constexpr (AKC) {
            const int64_t gm = m0 + a_row[i];
            a_off[i] = (uint32_t)(((gm < g.M ? gm : 0) * g.lda + a_c4[i] * 4) * 4);
        } else {
            const int64_t gm = m0 + a_c4[i] * 4;       // VEC: M % 4 == 0, a float4 is wholly in or out of range
            a_off[i] = (uint32_t)((a_row[i] * g.lda + (gm < g.M ? gm : 0)) * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int f = tid + i * NT;
        b_row[i] = BKC ? f / (BK / 4) : f / (BN / 4);
        b_c4[i] = BKC ? f % (BK / 4) : f % (BN / 4);
        b_dst[i] = b_row[i] * SB + b_c4[i] * 4;
        if constexpr (BKC) {
            const int64_t gn = n0 + b_row[i];
            b_off[i] = (uint32_t)(((gn < g.N ? gn : 0) * g.ldb + b_c4[i] * 4) * 4);
        } else {
            const int64_t gn = n0 + b_c4[i] * 4;
            b_off[i] = (uint32_t)((b_row[i] * g.ldb + (gn < g.N ? gn : 0)) * 4);
        }
    }
    const int64_t a_step = (AKC ? (int64_t)BK : (int64_t)BK * g.lda) * 4;   // bytes per K-tile
    const int64_t b_step = (BKC ? (int64_t)BK : (int64_t)BK * g.ldb) * 4;
    const char* const a_base = reinterpret_cast<const char*>(g.A) + (AKC ? kbeg : kbeg * g.lda) * 4;
    const char* const b_base = reinterpret_cast<const char*>(g.B) + (BKC ? kbeg : kbeg * g.ldb) * 4;

    float4 ra[A_F4], rb[B_F4];

    // fast loader: full K-tile, 16-B loads, no branches and NO use of the loaded value (a select here would
    // put an s_waitcnt right behind the loads and expose the whole global latency before the MFMAs).  Rows /
    // columns beyond M / N read a clamped, valid address: what they load only ever reaches accumulator rows /
    // columns >= M / N, which the epilogue never stores (an output row depends on its own A row only).
    // buffer loads: resource (4 SGPRs) + constant per-thread VGPR offset + a scalar offset that advances per K-tile —
    // no vector address arithmetic at all in the loop, and a single address VGPR per load
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(a_base), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(b_base), 0, 0xffffffffu, 0x00020000);
    auto load_full = [&](int kt) {
        const uint32_t at = (uint32_t)kt * (uint32_t)a_step, bt = (uint32_t)kt * (uint32_t)b_step;   // wave-uniform
#pragma unroll
        for (int i = 0; i < A_F4; ++i)
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_off[i], at, 0));
#pragma unroll
        for (int i = 0; i < B_F4; ++i)
            rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_off[i], bt, 0));
    };
    // guarded loader: element-wise bounds on every axis (K tail, odd shapes, unaligned operands)
    auto load_guarded = [&](int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * BK;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (AKC) {
                const int64_t gm = m0 + a_row[i], gk = k0 + a_c4[i] * 4;
                if (gm < g.M) {
                    const float* p = g.A + gm * g.lda + gk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (gk + j < kend) e[j] = p[j];
                }
            } else {
                const int64_t gk = k0 + a_row[i], gm = m0 + a_c4[i] * 4;
                if (gk < kend) {
                    const float* p = g.A + gk * g.lda + gm;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (gm + j < g.M) e[j] = p[j];
                }
            }
            ra[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BKC) {
                const int64_t gn = n0 + b_row[i], gk = k0 + b_c4[i] * 4;
                if (gn < g.N) {
                    const float* p = g.B + gn * g.ldb + gk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (gk + j < kend) e[j] = p[j];
                }
            } else {
                const int64_t gk = k0 + b_row[i], gn = n0 + b_c4[i] * 4;
                if (gk < kend) {
                    const float* p = g.B + gk * g.ldb + gn;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (gn + j < g.N) e[j] = p[j];
                }
            }
            rb[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
    };
    auto load_tile = [&](int kt) {
        if (VEC && kt < nk_full) load_full(kt);
        else load_guarded(kt);
    };

    auto store_tile = [&](int buf) {
        float* As = lds + buf * (A_ELEMS + B_ELEMS);
        float* Bs = As + A_ELEMS;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4*>(As + a_dst[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) *reinterpret_cast<float4*>(Bs + b_dst[i]) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS fragment base offsets (floats), fixed for the whole loop
    const int a_frag = AKC ? (wm * TM + l31) * SA + lhi * 4 : (lhi * 4) * SA + wm * TM + l31;
    const int b_frag = BKC ? (wn * TN + l31) * SB + lhi * 4 : (lhi * 4) * SB + wn * TN + l31;
    constexpr int KK = BK / 8;

    // Fragment registers: one slot per 8-deep chunk of the K-tile, filled TWO chunks ahead of their MFMAs.
    float af[KK][MI][4], bf[KK][NI][4];
    auto read_frag = [&](int buf, int kk) {
        const float* As = lds + buf * (A_ELEMS + B_ELEMS);
        const float* Bs = As + A_ELEMS;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if constexpr (AKC) {
                float4 v = *reinterpret_cast<const float4*>(As + a_frag + i * 32 * SA + kk * 8);
                af[kk][i][0] = v.x; af[kk][i][1] = v.y; af[kk][i][2] = v.z; af[kk][i][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) af[kk][i][j] = As[a_frag + (kk * 8 + j) * SA + i * 32];
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if constexpr (BKC) {
                float4 v = *reinterpret_cast<const float4*>(Bs + b_frag + i * 32 * SB + kk * 8);
                bf[kk][i][0] = v.x; bf[kk][i][1] = v.y; bf[kk][i][2] = v.z; bf[kk][i][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[kk][i][j] = Bs[b_frag + (kk * 8 + j) * SB + i * 32];
            }
        }
    };
    auto mfma_chunk = [&](int kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    // operands SWAPPED: the block accumulates (A B)^T, i.e. lane l31 owns output ROW l31 of the block and
                    // its registers (r & 3) hold 4 CONSECUTIVE COLUMNS — the epilogue leaves through 16-B stores
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[kk][ni][j], af[kk][mi][j],
                                                                       acc[mi][ni], 0, 0, 0);
    };

    // Software pipeline — one barrier per K-tile and no exposed LDS or global latency in steady state.
    // Chunk c runs the MFMAs of fragment c and, BEFORE them, issues the LDS reads of the fragment used two
    // chunks later (sched_barrier(0) pins "memory ops first, then the chunk's MFMAs": left alone hipcc sinks
    // the reads below the MFMAs and then drains them in front of the barrier):
    //   chunk 0 : write tile kt+1 to the other buffer (its global loads were issued a whole tile ago),
    //             issue the global loads of tile kt+2, read F(kt,2) and F(kt,3)           | MFMA F(kt,0)
    //   chunk 1 :                                                                         | MFMA F(kt,1)
    //   ---- s_barrier ---- (pinned, see pinned_barrier)
    //   chunk 2 : read F(kt+1,0), F(kt+1,1) from the other buffer                         | MFMA F(kt,2)
    //   chunk 3 :                                                                         | MFMA F(kt,3)
    // (the tail loop keeps the older placement: F(kt,3) in chunk 1, F(kt+1,1) in chunk 3 — same register slots)
    // Hazards: the buffer written in chunk 0 of tile kt was last read in chunks 0-1 of tile kt-1, i.e. before
    // the barrier of tile kt-1 that every wave has passed; it is first read after the barrier of tile kt,
    // which every wave reaches after its own writes.
    static_assert(KK == 4, "the pipeline below is written for BK = 32");
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

    // Interleave pattern for one chunk: each MFMA (64 cycles on the pipe) is followed by a few of the chunk's
    // memory / address instructions, so that they issue in the MFMA's shadow instead of in a burst between MFMA
    // groups (masks: 0x8 MFMA, 0x2 VALU, 0x4 SALU, 0x20 VMEM read, 0x100 DS read, 0x200 DS write).
    constexpr int NMF = MI * NI * 4;           // MFMAs per chunk
#define TNN_INTERLEAVE_LIGHT()                                             \
    _Pragma("unroll") for (int q = 0; q < NMF; ++q) {                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 \
    }
#define TNN_INTERLEAVE_HEAVY()                                             \
    _Pragma("unroll") for (int q = 0; q < NMF; ++q) {                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 \
    }

    // The barrier as the compiler must see it: hipcc hoists the register-only MFMAs of chunks 2-3 above a plain
    // __syncthreads() (nothing orders them against it), which leaves the post-barrier LDS reads with no MFMAs to hide
    // behind.  Empty asm statements that "redefine" the fragments of chunks 2-3 right after the barrier tie those
    // MFMAs to it (volatile asm statements keep their order).  lgkmcnt(0): this wave's tile stores have landed.
    auto pinned_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int kk = 2; kk < 4; ++kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(af[kk][i][j]));
#pragma unroll
                for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(bf[kk][i][j]));
            }
        }
    };
    // steady state: tiles kt+1 and kt+2 exist and tile kt+2 is a full, vector-loadable tile -> no branches.
    // Two tiles per trip with the buffer index a compile-time constant: every LDS address is then a loop-invariant
    // register plus an immediate offset (no per-tile address arithmetic next to the MFMAs).
    TNN_TRACE(1);
    int kt = 0;
    const int n_steady = VEC ? (nk_full - 2 < nk - 2 ? nk_full - 2 : nk - 2) : 0;
#define TNN_STEADY_TILE(CUR, KT)                                           \
    store_tile((CUR) ^ 1);                                                 \
    load_full((KT) + 2);                                                   \
    read_frag((CUR), 2);                                                   \
    read_frag((CUR), 3);                                                   \
    mfma_chunk(0);                                                         \
    TNN_INTERLEAVE_HEAVY();                                                \
    __builtin_amdgcn_sched_barrier(0);                                     \
    mfma_chunk(1);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                     \
    pinned_barrier();                                                      \
    __builtin_amdgcn_sched_barrier(0);                                     \
    read_frag((CUR) ^ 1, 0);                                               \
    read_frag((CUR) ^ 1, 1);                                               \
    mfma_chunk(2);                                                         \
    TNN_INTERLEAVE_LIGHT();                                                \
    __builtin_amdgcn_sched_barrier(0);                                     \
    mfma_chunk(3);                                                         \
    __builtin_amdgcn_sched_barrier(0);
    for (; kt + 1 < n_steady; kt += 2) {
        TNN_STEADY_TILE(0, kt)
        TNN_STEADY_TILE(1, kt + 1)
    }
#undef TNN_STEADY_TILE
#undef TNN_INTERLEAVE_LIGHT
#undef TNN_INTERLEAVE_HEAVY

    // remaining tiles (the last two, a partial K tail, or every tile of the guarded variant): same pipeline
    // with its conditionals
    for (; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
        if (has1) store_tile(cur ^ 1);
        if (has2) load_tile(kt + 2);
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

    TNN_TRACE(2);
    // epilogue: with the swapped operands lane l31 holds ROW l31 of each 32x32 block and register r the column
    // (r & 3) + 8 (r >> 2) + 4 lhi: four consecutive columns per register quad -> one 16-B store per quad (4 store
    // instructions per block instead of 16 dword stores; 8 instead of 32 per wave for the 128x64 tile)
    if (g.epi == EPI_ADAM) {
        // The tile is a weight gradient: Adam (core/optimizer.py:67-79, the maths of adam_kernel in tnn_fused.hip) updates the
        // matching block of p / m / v here, 16 B per lane and array, and the gradient goes to C only if the caller wants it
        // stored.  The fp32 product is MFMA-bound, so the optimizer's 24 B per parameter ride on an otherwise idle memory
        // system (host side guarantees: splits == 1, 16-B aligned arrays, ldc % 4 == 0, N % 4 == 0).
        if (g.ad_guard != nullptr && *g.ad_guard != 0) return;
        const float ic1 = (float)(1.0 / (1.0 - g.ad_pows[0])), ic2 = (float)(1.0 / (1.0 - g.ad_pows[1]));
        const float omb1 = 1.f - g.ad_b1, omb2 = 1.f - g.ad_b2, lr = g.ad_lr, eps = g.ad_eps;
        if (!BKC && g.cs_db != nullptr && blockIdx.z == 0 && (int)(m0 / BM) == (int)(n0 / BN) % g.tiles_m) {
            // ONE workgroup per tile column (tile row = column index mod tiles_m: spread over the XCDs and over the launch's
            // duration — with all of them in tile row 0 they sat on two XCDs) also produces db = column sums of B (= dz,
            // MN-contiguous: the TN form) for its BN columns and applies
            // Adam to the bias block (core/ops.py:52-54 + core/optimizer.py:67-79).  The K x BN panel was streamed through
            // this workgroup a moment ago (L2 hits); summing it again HERE, outside the K loop, costs the 64 workgroups of the
            // row a few microseconds and leaves the MFMA loop alone (a first version added f64 adds on the B fragments
            // inside the loop's wave-uniform branch: the dW launch went from 162 to 182 us — the interleave pattern of the
            // loop does not survive extra VALU work).  f64 accumulation like every reduction of the library (tnn_reduce).
            constexpr int C4 = BN / 4, PARTS = NT / C4;            // 16-B loads: thread = (4 columns, one of PARTS row classes)
            // the partial sums meet in the operand buffers (free after the K loop): NO LDS of their own — 8 KB more per
            // workgroup took the kernel from 3 to 2 workgroups per CU, for every product of the library (measured: +3 %)
            static_assert(NT % C4 == 0 && PARTS * BN * 8 <= 2 * (A_ELEMS + B_ELEMS) * 4, "column-sum lanes / LDS");
            __syncthreads();                                    // every wave is done with the last tile's fragments
            double (*cs_part)[BN] = reinterpret_cast<double (*)[BN]>(lds);
            const int c4 = tid % C4, part4 = tid / C4;
            double a4[4] = {0.0, 0.0, 0.0, 0.0};
            if (n0 + 4 * c4 + 3 < g.N && g.vecB) {
                const float* bp = g.B + n0 + 4 * c4;
                constexpr int UN = 8;                               // independent loads in flight per thread
                for (int64_t k0 = part4; k0 < g.K; k0 += (int64_t)PARTS * UN) {
                    f32x4 v[UN];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int64_t k = k0 + (int64_t)u * PARTS;
                        v[u] = k < g.K ? *reinterpret_cast<const f32x4*>(bp + k * g.ldb) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) a4[e] += (double)v[u][e];
                }
            } else {
                for (int e = 0; e < 4; ++e) {
                    const int64_t gc = n0 + 4 * c4 + e;
                    if (gc < g.N)
                        for (int64_t k = part4; k < g.K; k += PARTS) a4[e] += (double)g.B[k * g.ldb + gc];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) cs_part[part4][4 * c4 + e] = a4[e];
            __syncthreads();
            const int col = tid % BN, part = tid / BN;
            const int64_t gcol = n0 + col;
            if (part == 0 && gcol < g.N) {
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < PARTS; ++q) t += cs_part[q][col];
                const float sum = (float)t;
                g.cs_db[gcol] = sum;
                if (g.cs_p != nullptr) {
                    float mi = g.cs_m[gcol], vi = g.cs_v[gcol];
                    mi = mi + omb1 * (sum - mi);
                    vi = vi + omb2 * (sum * sum - vi);
                    g.cs_m[gcol] = mi;
                    g.cs_v[gcol] = vi;
                    g.cs_p[gcol] = g.cs_p[gcol] + (-lr * (mi * ic1) / (sqrtf(vi * ic2) + eps));
                }
            }
            __syncthreads();
        }
        constexpr int TS = BN + 4;                                    // LDS row stride of the staged tile (floats)
        if constexpr (BM * TS <= 2 * (A_ELEMS + B_ELEMS) && (BN / 4) <= NT && NT % (BN / 4) == 0) {
            // Interior tiles leave through LDS.  In the accumulator layout a lane owns a ROW and 4 consecutive columns: a 16-B
            // access per lane touches 32 rows x 32 B per instruction — fine for the one store of a plain product, but the
            // optimizer adds three loads and three stores per element (measured: the step got SLOWER, 0.85 -> 1.00 ms, with
            // that pattern).  Staged row-major, a wave instruction covers 4 whole 256-B tile rows.
            if (m0 + BM <= g.M && n0 + BN <= g.N) {                   // block-uniform
                __syncthreads();                                      // every wave is done with the last K-tile's fragments
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<f32x4*>(lds + (wm * TM + mi * 32 + l31) * TS + wn * TN + ni * 32 + 8 * q + 4 * lhi) =
                                f32x4{acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                __syncthreads();
                constexpr int CPR = BN / 4, RPP = NT / CPR, PASSES = BM / RPP;     // 16-B pieces per row, rows per pass
                const int c4 = tid % CPR, r0 = tid / CPR;
#pragma unroll 1
                for (int pb = 0; pb < PASSES; pb += 4) {
                    f32x4 pv[4], mv[4], vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {                     // 12 x 16-B loads in flight per lane
                        const int64_t o = (m0 + r0 + (pb + u) * RPP) * g.ldc + n0 + 4 * c4;
                        pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.ad_p + o));
                        mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.ad_m + o));
                        vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.ad_v + o));
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int row = r0 + (pb + u) * RPP;
                        const int64_t o = (m0 + row) * g.ldc + n0 + 4 * c4;
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(lds + row * TS + 4 * c4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            mv[u][j] = mv[u][j] + omb1 * (gv[j] - mv[u][j]);
                            vv[u][j] = vv[u][j] + omb2 * (gv[j] * gv[j] - vv[u][j]);
                            pv[u][j] = pv[u][j] + (-lr * (mv[u][j] * ic1) / (sqrtf(vv[u][j] * ic2) + eps));
                        }
                        __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(g.ad_m + o));
                        __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(g.ad_v + o));
                        *reinterpret_cast<f32x4*>(g.ad_p + o) = pv[u];             // the next forward reads it
                        if (g.C != nullptr) *reinterpret_cast<f32x4*>(g.C + o) = gv;
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int64_t row = m0 + wm * TM + mi * 32 + l31;
                if (row >= g.M) continue;
                f32x4 pv[4], mv[4], vv[4];
                bool live[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {                      // 12 x 16-B loads in flight per lane
                    const int64_t col = n0 + wn * TN + ni * 32 + 8 * q + 4 * lhi;
                    live[q] = col < g.N;
                    const int64_t o = row * g.ldc + (live[q] ? col : 0);
                    pv[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.ad_p + o));
                    mv[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.ad_m + o));
                    vv[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.ad_v + o));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!live[q]) continue;
                    const int64_t o = row * g.ldc + n0 + wn * TN + ni * 32 + 8 * q + 4 * lhi;
                    f32x4 gv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gi = acc[mi][ni][4 * q + j];
                        gv[j] = gi;
                        mv[q][j] = mv[q][j] + omb1 * (gi - mv[q][j]);
                        vv[q][j] = vv[q][j] + omb2 * (gi * gi - vv[q][j]);
                        pv[q][j] = pv[q][j] + (-lr * (mv[q][j] * ic1) / (sqrtf(vv[q][j] * ic2) + eps));
                    }
                    __builtin_nontemporal_store(mv[q], reinterpret_cast<f32x4*>(g.ad_m + o));
                    __builtin_nontemporal_store(vv[q], reinterpret_cast<f32x4*>(g.ad_v + o));
                    *reinterpret_cast<f32x4*>(g.ad_p + o) = pv[q];                 // the next forward reads it
                    if (g.C != nullptr) *reinterpret_cast<f32x4*>(g.C + o) = gv;
                }
            }
        return;
    }
    {
        float* const dst = g.splits > 1 ? g.ws + (int64_t)blockIdx.z * g.M * g.N : g.C;
        const int64_t ldd = g.splits > 1 ? g.N : g.ldc;
        const bool vec_c = (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && ldd % 4 == 0;
        constexpr int TS = BN + 4;
        if constexpr (BM * TS <= 2 * (A_ELEMS + B_ELEMS) && (BN / 4) <= NT && NT % (BN / 4) == 0) {
            // interior tiles leave through LDS too: row-major 16-B stores, a wave instruction covering whole 256-B tile rows
            // instead of 32 rows x 32 B (measured, 128x64 tiles: dW 4096x4096x512 103-105 -> 108-109 TFLOP/s, the M = 512
            // products +1-3 %)
            if (g.staged_c && vec_c && m0 + BM <= g.M && n0 + BN <= g.N) {
                __syncthreads();
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<f32x4*>(lds + (wm * TM + mi * 32 + l31) * TS + wn * TN + ni * 32 + 8 * q + 4 * lhi) =
                                f32x4{acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                __syncthreads();
                constexpr int CPR = BN / 4, RPP = NT / CPR, PASSES = BM / RPP;
                const int c4 = tid % CPR, r0 = tid / CPR;
#pragma unroll
                for (int pb = 0; pb < PASSES; ++pb) {
                    const int row = r0 + pb * RPP;
                    f32x4 gv = *reinterpret_cast<const f32x4*>(lds + row * TS + 4 * c4);
                    if (g.splits == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) gv[j] = apply_epilogue(g, gv[j], m0 + row, n0 + 4 * c4 + j);
                    }
                    *reinterpret_cast<f32x4*>(dst + (m0 + row) * ldd + n0 + 4 * c4) = gv;
                }
                return;
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int64_t row = m0 + wm * TM + mi * 32 + l31;
                if (row >= g.M) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t col = n0 + wn * TN + ni * 32 + 8 * q + 4 * lhi;
                    if (col >= g.N) continue;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[mi][ni][4 * q + j];
                        if (g.splits == 1 && col + j < g.N) v[j] = apply_epilogue(g, v[j], row, col + j);
                    }
                    if (vec_c && col + 3 < g.N) {
                        *reinterpret_cast<float4*>(dst + row * ldd + col) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (col + j < g.N) dst[row * ldd + col + j] = v[j];
                    }
                }
            }
    }
#ifdef TNN_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TNN_TRACE(3);
#endif
#undef TNN_TRACE
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
    int64_t total = g.M * g.N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < g.splits; ++z) s += g.ws[(int64_t)z * total + i];
        int64_t row = i / g.N, col = i - row * g.N;
        g.C[row * g.ldc + col] = apply_epilogue(g, s, row, col);
    }
}


// ------------------------------------------------------------------------------ small / latency GEMM
// MNIST-size layers (2MNK <= ~160 MFLOP) are latency-bound: the 64x64 LDS-tiled kernel needs split-K
// plus a second launch and still leaves most CUs idle.  Here one workgroup owns ONE 16x16 output tile
// (v_mfma_f32_16x16x4_f32, 4 accumulator VGPRs) and its WAVES waves split K between them in 16-deep
// chunks (round-robin, so neighbouring waves touch neighbouring lines); fragments are loaded straight
// from global memory in MFMA layout (everything is L2-resident at this size, an LDS round trip would
// only add latency), the WAVES partial tiles are summed through LDS and the epilogue is applied once.
// Optionally the workgroups of the first tile row also emit colsum[n] = sum_k B[k][n] — the bias
// gradient (core/ops.py:52-54) — from the B fragments they already hold (B = dZ in dW = X^T dZ).
//   lane l: i = l & 15 (row of A / column of B), grp = l >> 4 holds k = 16c + 4 grp + j, j = 0..3
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Optional optimizer tail of a dW tile (single-GPU step): the tile's 256 gradient values are in registers when the
// epilogue stores them, so Adam (core/optimizer.py:67-79, the maths of adam_kernel in tnn_fused.hip) is applied to
// the matching weights on the spot, and to the bias entries by the column-sum threads; pows already holds b1^t, b2^t
// of this step.  The parameter / moment values are requested before the K loop.
struct AdamEpi {
    float* pw; float* mw; float* vw;     // weight block [M, ldc] matching C
    float* pb; float* mb; float* vb;     // bias block [N] matching colsum
    float* fp; const float* fg; float* fm; float* fv; int64_t fn;   // extra flat range updated by the trailing blocks
    float lr, b1, b2, eps;
    const double* pows;
};
__device__ __forceinline__ float adam_apply(const AdamEpi& ad, float ic1, float ic2, float g, float& m, float& v, float p) {
    m = m + (1.f - ad.b1) * (g - m);
    v = v + (1.f - ad.b2) * (g * g - v);
    return p + (-ad.lr * (m * ic1) / (sqrtf(v * ic2) + ad.eps));
}

// TO_LDS: the finished tile goes to out_lds [16][16] (row-major; entries outside the matrix are not written) and the column
// sums of the first tile row to out_lds[256 .. 272) instead of memory — for a caller that sends them elsewhere itself
#ifdef TNN_STEP_TRACE
__device__ unsigned long long g_step_trace[4 * 1024 * 4];        // tnn_internal.h: TNN_STEP_STAMP
#endif

template <bool AKC, bool BKC, int WAVES, bool ADAM = false, bool TO_LDS = false>
__device__ __forceinline__ void small_tile(const GemmArgs& g, float* __restrict__ colsum, int block,
                                           float (*red)[4][64], float (*bsum)[64], const AdamEpi* ad = nullptr,
                                           float* out_lds = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i16 = lane & 15, grp = lane >> 4;
    int tm, tn;
    if (g.xg_m) {                                    // XCD-aware tile order (GemmArgs::xg_m)
        const int xcd = block & 7, idx = block >> 3, pm = g.tiles_m / g.xg_m, pn = g.tiles_n / g.xg_n;
        tm = (xcd % g.xg_m) * pm + idx % pm;
        tn = (xcd / g.xg_m) * pn + idx / pm;
    } else {
        tm = block % g.tiles_m;
        tn = block / g.tiles_m;
    }
    const int64_t m0 = (int64_t)tm * 16, n0 = (int64_t)tn * 16;
    const int64_t am = m0 + i16, bn = n0 + i16;
    const bool a_ok = am < g.M, b_ok = bn < g.N;
    const int nchunks = (int)((g.K + 15) / 16);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bs = 0.f;
    float a_p = 0.f, a_m = 0.f, a_v = 0.f, ab_p = 0.f, ab_m = 0.f, ab_v = 0.f, ic1 = 0.f, ic2 = 0.f;
    if constexpr (ADAM) TNN_STEP_STAMP(g_step_trace, 3, 0);
    if constexpr (ADAM) {
        ic1 = (float)(1.0 / (1.0 - ad->pows[0]));
        ic2 = (float)(1.0 / (1.0 - ad->pows[1]));
        if (tid < 256) {
            const int64_t row = m0 + ((tid & 63) >> 4) * 4 + (tid >> 6), col = n0 + (tid & 15);
            if (row < g.M && col < g.N) {
                const int64_t i = row * g.ldc + col;
                a_p = ad->pw[i]; a_m = ad->mw[i]; a_v = ad->vw[i];
            }
        }
        if (tm == 0 && tid < 16 && n0 + tid < g.N) {
            ab_p = ad->pb[n0 + tid]; ab_m = ad->mb[n0 + tid]; ab_v = ad->vb[n0 + tid];
        }
    }
    for (int c = wid; c < nchunks; c += WAVES) {
        const int64_t k = (int64_t)c * 16 + grp * 4;
        float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (AKC) {
            const float* p = g.A + am * g.lda + k;
            if (a_ok) {
                if (g.vecA && k + 3 < g.K) {
                    float4 v = *reinterpret_cast<const float4*>(p);
                    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < g.K) a[j] = p[j];
                }
            }
        } else {
            if (a_ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k + j < g.K) a[j] = g.A[(k + j) * g.lda + am];
            }
        }
        if constexpr (BKC) {
            const float* p = g.B + bn * g.ldb + k;
            if (b_ok) {
                if (g.vecB && k + 3 < g.K) {
                    float4 v = *reinterpret_cast<const float4*>(p);
                    b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < g.K) b[j] = p[j];
                }
            }
        } else {
            if (b_ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k + j < g.K) b[j] = g.B[(k + j) * g.ldb + bn];
            }
        }
        if constexpr (ADAM) { if (c == wid) TNN_STEP_STAMP_ACKED(g_step_trace, 3, 1); }      // (first chunk's operands + the Adam state)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
        bs += (b[0] + b[1]) + (b[2] + b[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][r][lane] = acc[r];
    bsum[wid][lane] = bs;
    __syncthreads();
    if constexpr (ADAM) TNN_STEP_STAMP(g_step_trace, 3, 2);
    if (tid < 256) {
        const int r = tid >> 6, ln = tid & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w][r][ln];
        const int64_t row = m0 + (ln >> 4) * 4 + r, col = n0 + (ln & 15);   // 16x16x4 C/D layout
        if (row < g.M && col < g.N) {
            const float gval = apply_epilogue(g, s, row, col);
            if constexpr (TO_LDS) out_lds[((ln >> 4) * 4 + r) * 16 + (ln & 15)] = gval;
            else if (!ADAM || g.C != nullptr) g.C[row * g.ldc + col] = gval;      // ADAM: the gradient itself only on request
            if constexpr (ADAM) {
                const int64_t i = row * g.ldc + col;
                ad->pw[i] = adam_apply(*ad, ic1, ic2, gval, a_m, a_v, a_p);
                ad->mw[i] = a_m;
                ad->vw[i] = a_v;
            }
        }
    }
    if ((TO_LDS || colsum != nullptr) && tm == 0 && tid < 16 && n0 + tid < g.N) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += (bsum[w][tid] + bsum[w][16 + tid]) + (bsum[w][32 + tid] + bsum[w][48 + tid]);
        if constexpr (TO_LDS) out_lds[256 + tid] = s;
        else colsum[n0 + tid] = s;
        if constexpr (ADAM) {
            ad->pb[n0 + tid] = adam_apply(*ad, ic1, ic2, s, ab_m, ab_v, ab_p);
            ad->mb[n0 + tid] = ab_m;
            ad->vb[n0 + tid] = ab_v;
        }
    }
    if constexpr (ADAM) TNN_STEP_STAMP_ACKED(g_step_trace, 3, 3);
}

// the same epilogue with its operand (bias value / mask source / old C) already in a register
__device__ __forceinline__ float finish_epilogue(const GemmArgs& g, float acc, float pre) {
    if (g.epi == EPI_AXPBY) {
        float r = g.alpha * acc;
        if (g.beta != 0.f) r += g.beta * pre;
        return r;
    } else if (g.epi == EPI_BIAS_ACT) {
        float v = acc + pre;
        if (g.act == TNN_ACT_RELU) {
            if (g.relu_sign) v = v < 0.f ? -0.0f : fabsf(v);
            else v = v < 0.f ? 0.f : v;
        }
        return v;
    } else {
        return (__float_as_uint(pre) >> 31) ? 0.f : acc;
    }
}

// FAST variant of small_tile for aligned operands (vecA && vecB, every extent below 4 GiB) — same tile, same lane
// layout, same reduction, but nothing on the path from launch to the first MFMA except the loads themselves:
//   * fragments come through buffer loads (SGPR resource + 32-bit offset): an out-of-range row / column / k gets the
//     offset 0xffffffff and the hardware returns 0 — no exec-mask branches, no 64-bit pointer arithmetic;
//   * a wave issues the loads of ALL its K-chunks (up to MAXC = 4 per batch) before the first MFMA; one chunk per
//     loop trip made every trip a dependent L2/HBM round trip (fwd0 of the MNIST net: 4 trips per wave);
//   * the epilogue's operand (bias / ReLU-mask source / old C) is requested up front by the threads that apply it,
//     instead of after the cross-wave reduction.
// SYS_HEADZ: the partial logits leave through write-through (system-scope) stores, because a workgroup of THIS launch
// on another XCD reads them back (dense_fwd_head_kernel)
// tile of workgroup `block` of the latency kernels: one rectangle of the tile grid per XCD (pick_xcd_cut) or column-major
__device__ __forceinline__ void small_tile_coords(const GemmArgs& g, const int block, int& tm, int& tn) {
    if (g.xg_m) {
        const int xcd = block & 7, idx = block >> 3, pm = g.tiles_m / g.xg_m, pn = g.tiles_n / g.xg_n;
        tm = (xcd % g.xg_m) * pm + idx % pm;
        tn = (xcd / g.xg_m) * pn + idx / pm;
    } else {
        tm = block % g.tiles_m;
        tn = block / g.tiles_m;
    }
}

// The first layer's weight gradient on 16 x 32 tiles (round 6): dW0 = X^T dZ of the MNIST net is 49 x 16 = 784 tiles of 16 x 16 —
// 3.06 per CU, and the 16 workgroups that are a CU's FOURTH end 1.3 us behind the rest (profiles/r06_stepA_stamps.txt: 4.16 against
// 2.85 / 3.04 / 3.55 us for the first / second / third 256).  Two column tiles per workgroup — ONE A fragment, two B fragments, two
// accumulators per wave — make it 392 equal workgroups, at most two per CU, with 12 instead of 16 operand loads per 512 outputs.
// TN form only (A [K][lda] and B [K][ldb], both MN-contiguous), EPI_AXPBY with beta = 0 (a gradient), N a multiple of 32.
//   lane (i16, grp): a[j] = A[16 c + 4 grp + j][m0 + i16], b0 / b1[j] = B[.][n0 + i16] / B[.][n0 + 16 + i16]
//   acc0 / acc1[r] = C[m0 + 4 grp + r][n0 + i16] / [.. + 16]   (16x16x4 C/D layout)
// ADAM / TO_LDS as in small_tile; out_lds: [16][32] row-major, then the 32 column sums.
template <int WAVES, bool ADAM, bool TO_LDS>
__device__ __forceinline__ void dw_tile_wide(const GemmArgs& g, float* __restrict__ colsum, const int block, float (*red)[8][64],
                                             float (*bsum)[2][64], const AdamEpi* ad = nullptr, float* out_lds = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i16 = lane & 15, grp = lane >> 4;
    int tm, tn;
    small_tile_coords(g, block, tm, tn);                     // g.tiles_n counts 32-column tiles here
    const int64_t m0 = (int64_t)tm * 16, n0 = (int64_t)tn * 32;
    const int64_t am = m0 + i16, bn = n0 + i16;
    const bool a_ok = am < g.M;
    const int nchunks = (int)((g.K + 15) / 16);
    // this thread's two outputs of the epilogue (threads < 256): rows m0 + 4 (ln >> 4) + r, columns n0 + (ln & 15) and + 16
    const int e_r = tid >> 6, e_ln = tid & 63;
    const int64_t e_row = m0 + (e_ln >> 4) * 4 + e_r, e_col = n0 + (e_ln & 15);
    const bool e_live = tid < 256 && e_row < g.M;
    float a_p[2] = {0.f, 0.f}, a_m[2] = {0.f, 0.f}, a_v[2] = {0.f, 0.f}, ab_p = 0.f, ab_m = 0.f, ab_v = 0.f, ic1 = 0.f, ic2 = 0.f;
    if constexpr (ADAM) {
        ic1 = (float)(1.0 / (1.0 - ad->pows[0]));
        ic2 = (float)(1.0 / (1.0 - ad->pows[1]));
        if (e_live) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t i = e_row * g.ldc + e_col + 16 * u;
                a_p[u] = ad->pw[i]; a_m[u] = ad->mw[i]; a_v[u] = ad->vw[i];
            }
        }
        if (tm == 0 && tid < 32) { ab_p = ad->pb[n0 + tid]; ab_m = ad->mb[n0 + tid]; ab_v = ad->vb[n0 + tid]; }
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bs0 = 0.f, bs1 = 0.f;
    if constexpr (ADAM) TNN_STEP_STAMP(g_step_trace, 3, 0);
    for (int c = wid; c < nchunks; c += WAVES) {
        const int64_t k = (int64_t)c * 16 + grp * 4;
        float a[4] = {0.f, 0.f, 0.f, 0.f}, b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k + j < g.K) {
                if (a_ok) a[j] = g.A[(k + j) * g.lda + am];
                b0[j] = g.B[(k + j) * g.ldb + bn];
                b1[j] = g.B[(k + j) * g.ldb + bn + 16];
            }
        }
        if constexpr (ADAM) { if (c == wid) TNN_STEP_STAMP_ACKED(g_step_trace, 3, 1); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc1, 0, 0, 0);
        }
        bs0 += (b0[0] + b0[1]) + (b0[2] + b0[3]);
        bs1 += (b1[0] + b1[1]) + (b1[2] + b1[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[wid][r][lane] = acc0[r]; red[wid][4 + r][lane] = acc1[r]; }
    bsum[wid][0][lane] = bs0;
    bsum[wid][1][lane] = bs1;
    __syncthreads();
    if constexpr (ADAM) TNN_STEP_STAMP(g_step_trace, 3, 2);
    if (e_live) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float sres = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) sres += red[w][4 * u + e_r][e_ln];
            const float gval = g.alpha * sres;
            const int64_t i = e_row * g.ldc + e_col + 16 * u;
            if constexpr (TO_LDS) out_lds[(int)(e_row - m0) * 32 + (int)(e_col - n0) + 16 * u] = gval;
            else if (!ADAM || g.C != nullptr) g.C[i] = gval;      // ADAM: the gradient itself only on request
            if constexpr (ADAM) {
                ad->pw[i] = adam_apply(*ad, ic1, ic2, gval, a_m[u], a_v[u], a_p[u]);
                ad->mw[i] = a_m[u];
                ad->vw[i] = a_v[u];
            }
        }
    }
    if ((TO_LDS || colsum != nullptr) && tm == 0 && tid < 32) {
        const int u = tid >> 4, t = tid & 15;
        float sres = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sres += (bsum[w][u][t] + bsum[w][u][16 + t]) + (bsum[w][u][32 + t] + bsum[w][u][48 + t]);
        if constexpr (TO_LDS) out_lds[512 + tid] = sres;
        else colsum[n0 + tid] = sres;
        if constexpr (ADAM) {
            ad->pb[n0 + tid] = adam_apply(*ad, ic1, ic2, sres, ab_m, ab_v, ab_p);
            ad->mb[n0 + tid] = ab_m;
            ad->vb[n0 + tid] = ab_v;
        }
    }
    if constexpr (ADAM) TNN_STEP_STAMP_ACKED(g_step_trace, 3, 3);
}

template <bool AKC, bool BKC, int WAVES, bool SYS_HEADZ = false>
__device__ __forceinline__ void small_tile_fast(const GemmArgs& g, float* __restrict__ colsum, int block,
                                                float (*red)[4][64], float (*bsum)[64], float* head_lds = nullptr) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    constexpr uint32_t OOB = 0xffffffffu;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i16 = lane & 15, grp = lane >> 4;
    int tm, tn;
    small_tile_coords(g, block, tm, tn);
    const int64_t m0 = (int64_t)tm * 16, n0 = (int64_t)tn * 16;
    const int64_t am = m0 + i16, bn = n0 + i16;
    const bool a_ok = am < g.M, b_ok = bn < g.N;
    const int nchunks = (int)((g.K + 15) / 16);
    const uint32_t K = (uint32_t)g.K, lda4 = (uint32_t)g.lda * 4u, ldb4 = (uint32_t)g.ldb * 4u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A), 0, (uint32_t)(((AKC ? g.M : g.K) - 1) * g.lda + (AKC ? g.K : g.M)) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.B), 0, (uint32_t)(((BKC ? g.N : g.K) - 1) * g.ldb + (BKC ? g.K : g.N)) * 4u, 0x00020000);
    // byte offset of this lane's first element of chunk 0 (k = grp * 4), without the chunk term
    const uint32_t a_lane = AKC ? (uint32_t)am * lda4 + (uint32_t)grp * 16u : (uint32_t)grp * 4u * lda4 + (uint32_t)am * 4u;
    const uint32_t b_lane = BKC ? (uint32_t)bn * ldb4 + (uint32_t)grp * 16u : (uint32_t)grp * 4u * ldb4 + (uint32_t)bn * 4u;
    const uint32_t a_chunk = AKC ? 64u : 16u * lda4;          // bytes per 16-deep chunk
    const uint32_t b_chunk = BKC ? 64u : 16u * ldb4;

    const int e_r = tid >> 6, e_ln = tid & 63;
    const int64_t e_row = m0 + (e_ln >> 4) * 4 + e_r, e_col = n0 + (e_ln & 15);   // 16x16x4 C/D layout
    const bool e_live = tid < 256 && e_row < g.M && e_col < g.N;
    float e_pre = 0.f;
    if (e_live) {
        if (g.epi == EPI_BIAS_ACT) e_pre = g.bias ? g.bias[e_col] : 0.f;
        else if (g.epi == EPI_MASK) e_pre = g.Y[e_row * g.ldy + e_col];
        else if (g.beta != 0.f) e_pre = g.C[e_row * g.ldc + e_col];
    }
    float head_pre[4] = {0.f, 0.f, 0.f, 0.f};                 // wave 0: head_w[n0 + 4 grp + j][i16], the B fragment of the partial-logit product
    if (g.head_z != nullptr && wid == 0 && i16 < g.head_c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + 4 * grp + j < g.N) head_pre[j] = g.head_w[(n0 + 4 * grp + j) * g.head_c + i16];
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bs = 0.f;
    constexpr int MAXC = 4;
    constexpr int KID = WAVES == 16 ? 0 : 1;              // (trace build: fwd0 runs 16 waves, fwd1 4 — or 8 in the data-parallel tail form)
    TNN_STEP_STAMP(g_step_trace, KID, 0);
    for (int c0 = wid; c0 < nchunks; c0 += WAVES * MAXC) {
        float a[MAXC][4], b[MAXC][4];
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            const uint32_t c = (uint32_t)(c0 + u * WAVES);
            const uint32_t k = c * 16u + (uint32_t)grp * 4u;
            if ((int)c >= nchunks) {                                  // wave-uniform: no loads for chunks that do not exist
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[u][j] = 0.f; b[u][j] = 0.f; }
                continue;
            }
            if constexpr (AKC) {
                const uint32_t off = (a_ok && k < K) ? a_lane + c * a_chunk : OOB;     // K % 4 == 0: all in or all out
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[u][j] = __uint_as_float(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t off = (a_ok && k + j < K) ? a_lane + c * a_chunk + (uint32_t)j * lda4 : OOB;
                    a[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off, 0, 0));
                }
            }
            if constexpr (BKC) {
                const uint32_t off = (b_ok && k < K) ? b_lane + c * b_chunk : OOB;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, off, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[u][j] = __uint_as_float(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t off = (b_ok && k + j < K) ? b_lane + c * b_chunk + (uint32_t)j * ldb4 : OOB;
                    b[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rsrc, off, 0, 0));
                }
            }
        }
        if (c0 == wid) TNN_STEP_STAMP_ACKED(g_step_trace, KID, 1);
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u][j], acc, 0, 0, 0);
            bs += (b[u][0] + b[u][1]) + (b[u][2] + b[u][3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][r][lane] = acc[r];
    bsum[wid][lane] = bs;
    __syncthreads();
    TNN_STEP_STAMP(g_step_trace, KID, 2);
    float e_val = 0.f;
    if (tid < 256) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w][e_r][e_ln];
        if (e_live) {
            e_val = finish_epilogue(g, s, e_pre);
            g.C[e_row * g.ldc + e_col] = e_val;
        }
    }
    if (colsum != nullptr && tm == 0 && tid < 16 && n0 + tid < g.N) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += (bsum[w][tid] + bsum[w][16 + tid]) + (bsum[w][32 + tid] + bsum[w][48 + tid]);
        colsum[n0 + tid] = s;
    }
    if (head_lds != nullptr && g.head_z != nullptr) {
        // This tile's share of the NEXT layer's logits: head_z[tn][row][c] = sum over the tile's 16 columns of
        // out[row][col] * head_w[n0 + col][c] (a sign-encoded ReLU zero, -0.0, contributes -0 * w = 0).  The classifier
        // head then only ADDS tiles_n partials per logit instead of re-reading the whole activation and redoing the
        // product in every workgroup (csrc/tnn_head.hip).  The finished 16 x 16 tile goes through LDS once and wave 0
        // multiplies it with head_w's 16 rows (fragments requested at kernel start) in four 16x16x4 MFMAs.
        float* tile_s = head_lds;                               // [16][20]: 16-B aligned rows, conflict-free 16-B reads
        if (tid < 256) tile_s[((e_ln >> 4) * 4 + e_r) * 20 + (e_ln & 15)] = e_val;
        __syncthreads();
        if (wid == 0) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(tile_s + i16 * 20 + 4 * grp);     // out[row i16][col 4 grp + j]
            f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], head_pre[j], hacc, 0, 0, 0);
            if (i16 < g.head_c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = m0 + 4 * grp + q;
                    if (row < g.M) {
                        float* dst = g.head_z + ((int64_t)tn * g.M + row) * g.head_c + i16;
                        if constexpr (SYS_HEADZ) tnn::p2p::store_sys(reinterpret_cast<uint32_t*>(dst), __float_as_uint(hacc[q]));
                        else *dst = hacc[q];
                    }
                }
            }
        }
    }
    TNN_STEP_STAMP_ACKED(g_step_trace, KID, 3);
}

template <bool AKC, bool BKC, int WAVES, bool FAST>
__global__ __launch_bounds__(WAVES * 64) void gemm_small_f32_kernel(GemmArgs g, float* __restrict__ colsum) {
    __shared__ float red[WAVES][4][64];
    __shared__ float bsum[WAVES][64];
    if constexpr (FAST) {
        __shared__ __attribute__((aligned(16))) float head_lds[16 * 20];   // tnn_dense_fwd_head_partials: the finished tile
        if (g.bump != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *g.bump += 1u;     // single writer; read by the NEXT launch
        small_tile_fast<AKC, BKC, WAVES>(g, colsum, (int)blockIdx.x, red, bsum, head_lds);
    } else {
        small_tile<AKC, BKC, WAVES>(g, colsum, (int)blockIdx.x, red, bsum);
    }
}

// Forward of the hidden layer in front of a classifier head in a DATA-PARALLEL step: the tiles + their partial logits as
// above, and the workgroup that finishes LAST (an agent-scope ticket drawn after its own partial logits were
// acknowledged) reduces the shard's whole-batch softmax statistics from all the partials and — on the peer-to-peer
// transport — exchanges and merges them with the other ranks' (head_tail_stats, tnn_head_stats.h).  The head launch
// behind this one only READS the pair(s): no statistics launch, no workgroup of the head waiting for a peer, no
// requirement that the ranks' launches be resident together (core/losses.py:26-27 is why the exchange exists).
#ifdef TNN_AR_TRACE
__device__ unsigned long long g_fh_trace[128 * 8];   // debug build: stamps of dense_fwd_head_kernel (tools/probes/ar_fused_trace.py)
#endif
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dense_fwd_head_kernel(GemmArgs g, HeadTail ta, tnn::p2p::LaunchCtx ctx) {
    static_assert(WAVES == 8, "head_tail_stats is written for 512 threads");
    __shared__ float red[WAVES][4][64];
    __shared__ float bsum[WAVES][64];
    __shared__ __attribute__((aligned(16))) float head_lds[16 * 20];
    __shared__ __attribute__((aligned(16))) float zs[128 * 10], ys[128 * 10];
    __shared__ double dred[8][4];
    __shared__ int is_last;
#ifdef TNN_AR_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 128) g_fh_trace[blockIdx.x * 8] = wall_clock64();
#endif
    small_tile_fast<true, false, WAVES, true>(g, nullptr, (int)blockIdx.x, red, bsum, head_lds);
#ifdef TNN_AR_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 128) g_fh_trace[blockIdx.x * 8 + 1] = wall_clock64();
#endif
    // Shards of more than 128 rows: every 128-row block has its own arrival counter (ticket[1 + rb], the block's 8 tile rows x
    // all tile columns), so the blocks' statistics are reduced in parallel, each by the workgroup that finishes the block; the
    // pairs meet through memory (system scope) behind a second counter (ticket[0]) whose last arrival merges them in block
    // order.  Up to 128 rows: one counter, one pair, as before.
    using namespace tnn::p2p;
    int tm, tn;
    small_tile_coords(g, (int)blockIdx.x, tm, tn);
    const int nb = (ta.m + 127) / 128, rb = nb > 1 ? tm >> 3 : 0;
    const unsigned cnt = nb > 1 ? (unsigned)(min(8, g.tiles_m - 8 * rb) * g.tiles_n) : gridDim.x;
    unsigned int* const ctr = ta.ticket + (nb > 1 ? 1 + rb : 0);
    float* const pairs = reinterpret_cast<float*>(ta.ticket + 16);          // [8][2] behind the 16 counters
    if (threadIdx.x < 64) {                                    // wave 0 wrote this tile's partial logits
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TNN_AR_TRACE
        if (threadIdx.x == 0 && blockIdx.x < 128) g_fh_trace[blockIdx.x * 8 + 2] = wall_clock64();
#endif
        if (threadIdx.x == 0) {
            const unsigned prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == cnt - 1 ? 1 : 0;
            if (last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // graph replays start from 0
            is_last = last;
        }
    }
    __syncthreads();
#ifdef TNN_AR_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 128) { g_fh_trace[blockIdx.x * 8 + 3] = wall_clock64(); g_fh_trace[blockIdx.x * 8 + 6] = is_last; }
#endif
    if (!is_last) return;
    float M, S;
    head_tail_stats<10, 8>(ta, 128 * rb, min(ta.m, 128 * rb + 128), zs, ys, dred, M, S);
#ifdef TNN_AR_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 128) g_fh_trace[blockIdx.x * 8 + 4] = wall_clock64();
#endif
    if (nb > 1) {
        __syncthreads();                                       // is_last is about to be reused
        if (threadIdx.x == 0) {
            __hip_atomic_store(pairs + 2 * rb, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pairs + 2 * rb + 1, S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned prev = __hip_atomic_fetch_add(ta.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == (unsigned)nb - 1 ? 1 : 0;
            if (last) __hip_atomic_store(ta.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            is_last = last;
        }
        __syncthreads();
        if (!is_last) return;
        M = -INFINITY; S = 0.f;
        for (int q = 0; q < nb; ++q) {                         // block order: the same sum whichever block finished last
            const float mq = __hip_atomic_load(pairs + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const float sq = __hip_atomic_load(pairs + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const float nm = fmaxf(M, mq);
            S = S * expf(M - nm) + sq * expf(mq - nm);
            M = nm;
        }
    }
    head_tail_finish(ta, ctx, M, S);
#ifdef TNN_AR_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 128) g_fh_trace[blockIdx.x * 8 + 5] = wall_clock64();
#endif
}

// The same launch for ANY classifier head the merged head + hidden-backward launch takes (tnn_mlp_head_bwd_fits: hidden width a
// multiple of 16 up to 256, <= 16 classes; e.g. the 32 -> 10 tail of the reference's own example net, examples/mnist/run.py:59-69)
// and up to 1024 rows: ONE arrival counter for the launch; the last workgroup to arrive reduces the shard's {max, sum-exp} with a
// thread per row (rows t, t + 512: logits = bias + the tiles' partials in tile order, row max, row sum-exp; then the rows' pairs
// are merged by DPP wave reductions and an 8-entry LDS exchange — a fixed order, so the pair does not depend on which workgroup
// came last).  The tuned 128 -> 10 form above keeps its per-128-row-block counters and four-threads-per-row statistics.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dense_fwd_head_generic_kernel(GemmArgs g, HeadTail ta, tnn::p2p::LaunchCtx ctx) {
    static_assert(WAVES == 8, "512 threads");
    __shared__ float red[WAVES][4][64];
    __shared__ float bsum[WAVES][64];
    __shared__ __attribute__((aligned(16))) float head_lds[16 * 20];
    __shared__ float wred[WAVES][2];
    __shared__ int is_last;
    small_tile_fast<true, false, WAVES, true>(g, nullptr, (int)blockIdx.x, red, bsum, head_lds);
    using namespace tnn::p2p;
    if (threadIdx.x < 64) {                                    // wave 0 wrote this tile's partial logits
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const unsigned prev = __hip_atomic_fetch_add(ta.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == gridDim.x - 1 ? 1 : 0;
            if (last) __hip_atomic_store(ta.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // graph replays start from 0
            is_last = last;
        }
    }
    __syncthreads();
    if (!is_last) return;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int C = g.head_c, NP = g.tiles_n, m = ta.m;
    float mx = -INFINITY, sx = 0.f;                            // this thread's rows: {max, sum-exp relative to it}
    for (int r = t; r < m; r += WAVES * 64) {
        uint32_t u[16][1];
        float z[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) z[k] = k < C ? ta.bias[k] : -INFINITY;
        for (int tn = 0; tn < NP; ++tn) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ta.zpart) + ((size_t)tn * m + r) * C;
#pragma unroll
            for (int k = 0; k < 16; ++k) if (k < C) load_sys(u[k][0], src + k);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                asm volatile("" : "+v"(u[k][0]));
                if (k < C) z[k] += __uint_as_float(u[k][0]);
            }
        }
        float rm = -INFINITY, rs = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < C) rm = fmaxf(rm, z[k]);
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < C) rs += expf(z[k] - rm);
        const float nm = fmaxf(mx, rm);
        sx = sx * expf(mx - nm) + rs * expf(rm - nm);          // (mx = -inf, sx = 0 on the first row: 0 * exp(-inf) = 0)
        mx = nm;
    }
    const float wm = tnn::wave_max_dpp(mx);
    const float ws = tnn::wave_sum_dpp(mx > -INFINITY ? sx * expf(mx - wm) : 0.f);
    if (lane == 0) { wred[wid][0] = wm; wred[wid][1] = ws; }
    __syncthreads();
    float M = -INFINITY, S = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) M = fmaxf(M, wred[w][0]);
#pragma unroll
    for (int w = 0; w < WAVES; ++w) S += wred[w][0] > -INFINITY ? wred[w][1] * expf(wred[w][0] - M) : 0.f;
    head_tail_finish(ta, ctx, M, S);
}

// Forward of the hidden layer in front of a classifier head for batches of MORE than 128 rows on one GPU, ROW-PANEL form:
// a workgroup owns 16 whole rows — its 8 waves take the 8 column tiles of the 128 hidden units, each over the whole K — so
// the classifier's logits of those rows (8 partial products summed through LDS) and the rows' softmax statistics are
// finished INSIDE the workgroup: no arrival counter, no system-scope re-read of partial logits at the tail of the launch
// (2.3 us per 128-row block in dense_fwd_head_kernel, measured).  Outputs: the activations (bias + ReLU, sign-encoded
// zeros), zfull [M][10] = the logits WITHOUT the classifier bias (the head launch adds it, as it does to summed partials),
// pairs [ceil(M / 16)][2] = {max, sum-exp relative to it} of each 16-row panel — the head launch merges them.
struct RowPanelArgs {
    const float *A, *B, *bias;       // A [M][lda] (K-contiguous), B [K][ldb] (128 columns), bias [128]
    float* C;                        // [M][ldc]
    int64_t lda, ldb, ldc;
    int M, K, relu_sign;
    const float *head_w, *head_b;    // [128][10], [10]
    float *zfull, *pairs;
    unsigned int* bump;              // !MERGE: launch sequence of the deferred statistics exchange (GemmArgs::bump), or NULL
};

// MERGE (data-parallel step, rows > 128): the panels' pairs also meet inside the launch — every workgroup stores its pair at
// system scope and draws a ticket; the LAST one merges the <= 64 pairs (DPP tree over the panel index: a fixed order), on the
// peer-to-peer transport exchanges the shard's pair with the other ranks', and leaves ONE pair in ta.out_pair for the head
// launch.  What the counter form (dense_fwd_head_kernel) does with a system-scope re-read of all partial logits per 128-row
// block (2.3 us each), this does with 16 floats per panel: 54.8 -> 45.5 us per step at 1024 rows on one GPU carried over to the
// sharded step.
template <bool MERGE>
__global__ __launch_bounds__(512) void dense_fwd_rowpanel_head_kernel(RowPanelArgs g, HeadTail ta, tnn::p2p::LaunchCtx ctx) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    constexpr uint32_t OOB = 0xffffffffu;
    constexpr int C = 10;
    __shared__ __attribute__((aligned(16))) float tile_s[8][16 * 20];     // each wave's finished 16 x 16 tile, 16-B aligned rows
    __shared__ float zred[8][16 * 16];                                    // partial logits [wave][row][class slot]
    __shared__ float wred[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i16 = lane & 15, grp = lane >> 4;
    const int m0 = 16 * (int)blockIdx.x, n0 = 16 * wid;
    if constexpr (!MERGE) {
        if (g.bump != nullptr && blockIdx.x == 0 && tid == 0) *g.bump += 1u;       // single writer; read by the NEXT launch
    }
    const uint32_t K = (uint32_t)g.K, lda4 = (uint32_t)g.lda * 4u, ldb4 = (uint32_t)g.ldb * 4u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A), 0, (uint32_t)(((int64_t)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.B), 0, (uint32_t)(((int64_t)(g.K - 1) * g.ldb + 128) * 4), 0x00020000);
    const uint32_t b_lane = (uint32_t)grp * 4u * ldb4 + (uint32_t)(n0 + i16) * 4u;  // + 16 rows per chunk

    // epilogue operands, requested first
    const float e_bias = g.bias ? g.bias[n0 + i16] : 0.f;
    float head_pre[4] = {0.f, 0.f, 0.f, 0.f};                 // head_w[n0 + 4 grp + j][class i16]: the B fragment of the logits product
    if (i16 < C) {
#pragma unroll
        for (int j = 0; j < 4; ++j) head_pre[j] = g.head_w[(n0 + 4 * grp + j) * C + i16];
    }
    const int zrow = tid >> 4, zcls = tid & 15;               // threads < 256: logit (row, class slot)
    const float zb = (tid < 256 && zcls < C) ? g.head_b[zcls] : 0.f;

    // K loop in super-blocks of 256: the 16 rows of A go through LDS once per workgroup (all eight waves need the same rows),
    // and a wave requests ALL its B fragments of the super-block before the first MFMA (64 dword loads in flight, one round
    // trip) — with 16 / 32 / 64 workgroups in the launch it is one workgroup's latency that the launch takes.
    constexpr int KB = 256, CH = KB / 16, AS = KB + 4;
    __shared__ __attribute__((aligned(16))) float a_s[16 * AS];
    f32x4 accq[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // independent MFMA chains
    for (int kb = 0; kb < g.K; kb += KB) {
        float b[CH][4];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint32_t k = (uint32_t)kb + (uint32_t)c * 16u + (uint32_t)grp * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t boff = (k + j < K) ? b_lane + (uint32_t)(kb / 16 + c) * 16u * ldb4 + (uint32_t)j * ldb4 : OOB;
                b[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rsrc, boff, 0, 0));
            }
        }
        // A[16 rows][kb .. kb + 256): 1024 pieces of 16 B, two per thread (rows beyond M / columns beyond K: zero-filled)
        u32x4_t av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = tid + i * 512, row = piece >> 6, c4 = piece & 63;
            const uint32_t k = (uint32_t)kb + (uint32_t)c4 * 4u;
            const uint32_t aoff = (m0 + row < g.M && k < K) ? (uint32_t)(m0 + row) * lda4 + k * 4u : OOB;
            av[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, aoff, 0, 0);
        }
        if (kb) __syncthreads();                              // the previous super-block's fragments have been read
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = tid + i * 512, row = piece >> 6, c4 = piece & 63;
            *reinterpret_cast<u32x4_t*>(a_s + row * AS + 4 * c4) = av[i];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(a_s + i16 * AS + 16 * c + 4 * grp);
#pragma unroll
            for (int j = 0; j < 4; ++j) accq[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b[c][j], accq[j], 0, 0, 0);
        }
    }
    const f32x4 acc = (accq[0] + accq[1]) + (accq[2] + accq[3]);

    // epilogue: acc[r] = out[row 4 grp + r][col i16] of this wave's tile (16x16x4 C/D layout)
    float* ts = tile_s[wid];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = acc[r] + e_bias;
        if (g.relu_sign) v = v < 0.f ? -0.0f : fabsf(v);
        else v = v < 0.f ? 0.f : v;
        const int row = m0 + 4 * grp + r;
        if (row < g.M) g.C[(int64_t)row * g.ldc + n0 + i16] = v;
        ts[(4 * grp + r) * 20 + i16] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        // this wave's share of the logits: out[16 rows][its 16 units] x head_w[those units][classes] (a sign-encoded zero,
        // -0.0, contributes -0 * w = 0)
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ts + i16 * 20 + 4 * grp);        // out[row i16][unit 4 grp + j]
        f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], head_pre[j], hacc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) zred[wid][(4 * grp + q) * 16 + i16] = hacc[q];          // [row 4 grp + q][class i16]
    }
    __syncthreads();
    if (tid < 256) {
        const float* zr = &zred[0][zrow * 16 + zcls];
        const float z_nb = ((zr[0] + zr[256]) + (zr[512] + zr[768])) + ((zr[1024] + zr[1280]) + (zr[1536] + zr[1792]));
        const int grow = m0 + zrow;
        const bool valid = zcls < C && grow < g.M;
        if (valid) g.zfull[(int64_t)grow * C + zcls] = z_nb;
        const float z = z_nb + zb;                            // what the head launch reconstructs: (sum of "partials") + bias
        // row statistics over the 16 lanes of a DPP row (quad swaps, half-row mirror, row mirror: every lane ends with the total)
        float mx = valid ? z : -INFINITY, w;
        w = tnn::dpp_move<0xB1, 0xf>(-INFINITY, mx); mx = w > mx ? w : mx;
        w = tnn::dpp_move<0x4E, 0xf>(-INFINITY, mx); mx = w > mx ? w : mx;
        w = tnn::dpp_move<0x141, 0xf>(-INFINITY, mx); mx = w > mx ? w : mx;
        w = tnn::dpp_move<0x140, 0xf>(-INFINITY, mx); mx = w > mx ? w : mx;
        float se = valid ? expf(z - mx) : 0.f;
        se += tnn::dpp_move<0xB1, 0xf>(0.f, se);
        se += tnn::dpp_move<0x4E, 0xf>(0.f, se);
        se += tnn::dpp_move<0x141, 0xf>(0.f, se);
        se += tnn::dpp_move<0x140, 0xf>(0.f, se);
        // the wave's four rows (lanes with class slot 0 speak for their row), then the four waves through LDS
        const bool speaks = zcls == 0 && grow < g.M;
        const float wm = tnn::wave_max_dpp(speaks ? mx : -INFINITY);
        const float ws = tnn::wave_sum_dpp(speaks ? se * expf(mx - wm) : 0.f);
        if (lane == 0) { wred[wid][0] = wm; wred[wid][1] = ws; }
    }
    __syncthreads();
    if (tid == 0) {
        float M = wred[0][0];
#pragma unroll
        for (int q = 1; q < 4; ++q) M = fmaxf(M, wred[q][0]);
        float S = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) S += wred[q][0] > -INFINITY ? wred[q][1] * expf(wred[q][0] - M) : 0.f;
        if constexpr (MERGE) {
            __hip_atomic_store(g.pairs + 2 * blockIdx.x, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(g.pairs + 2 * blockIdx.x + 1, S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            g.pairs[2 * blockIdx.x] = M;
            g.pairs[2 * blockIdx.x + 1] = S;
        }
    }
    if constexpr (MERGE) {
        __shared__ int is_last;
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned prev = __hip_atomic_fetch_add(ta.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == gridDim.x - 1 ? 1 : 0;
            if (last) __hip_atomic_store(ta.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // graph replays start from 0
            is_last = last;
        }
        __syncthreads();
        if (!is_last) return;
        const int n = (int)gridDim.x;                              // <= 64 panels (1024 rows)
        const bool has = lane < n;
        const float mq = has ? __hip_atomic_load(g.pairs + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : -INFINITY;
        const float sq = has ? __hip_atomic_load(g.pairs + 2 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
        const float Mm = tnn::wave_max_dpp(mq);                    // every wave computes the same pair
        const float Sm = tnn::wave_sum_dpp(has ? sq * expf(mq - Mm) : 0.f);
        head_tail_finish(ta, ctx, Mm, Sm);
    }
}

// Backward of one Dense layer in ONE launch: blocks [0, n_dw) compute dW = X^T dZ (TN) + db = colsum(dZ),
// blocks [n_dw, n_dw + n_dx) compute dX = (dZ W^T) * mask (NT, sign-bit mask epilogue).  The two products
// only share their input dZ, so they are independent grids fused to save a kernel boundary.
template <int WAVES, bool FAST>
__global__ __launch_bounds__(WAVES * 64) void dense_bwd_small_kernel(GemmArgs gw, float* __restrict__ db,
                                                                     GemmArgs gx, int n_dw) {
    __shared__ float red[WAVES][4][64];
    __shared__ float bsum[WAVES][64];
    if constexpr (FAST) {
        if ((int)blockIdx.x < n_dw)
            small_tile_fast<false, false, WAVES>(gw, db, (int)blockIdx.x, red, bsum);
        else
            small_tile_fast<true, true, WAVES>(gx, nullptr, (int)blockIdx.x - n_dw, red, bsum);
    } else {
        if ((int)blockIdx.x < n_dw)
            small_tile<false, false, WAVES>(gw, db, (int)blockIdx.x, red, bsum);
        else
            small_tile<true, true, WAVES>(gx, nullptr, (int)blockIdx.x - n_dw, red, bsum);
    }
}

// XCD-aware cut of a 16 x 16 tile grid (GemmArgs::xg_m / xg_n): the one with the smallest operand footprint per XCD
void pick_xcd_cut(GemmArgs& g) {
    g.xg_m = g.xg_n = 0;
    int best = 0;
    for (int xm = 1; xm <= 8; xm *= 2) {
        const int xn = 8 / xm;
        if (g.tiles_m % xm || g.tiles_n % xn) continue;
        const int cost = g.tiles_m / xm + g.tiles_n / xn;
        if (!best || cost < best) { best = cost; g.xg_m = xm; g.xg_n = xn; }
    }
}

// branch-free buffer-load variant: 16-B loads on the K-contiguous operands need alignment and K % 4 == 0; 32-bit
// offsets need every extent below 4 GiB (always true at this kernel's problem sizes, checked anyway)
bool small_fast_ok(const GemmArgs& g, int transA, int transB) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool akc = !transA, bkc = transB != 0;
    // measured on the MNIST layers: NN fwd1 3.41 -> 2.87 us, fwd0 4.50 -> 4.26; the TN products (both operands read
    // with 4-B loads either way) 3.98 -> 4.24 — they keep the plain loop
    if (!akc && !bkc) return false;
    if (akc && !(al(g.A) && g.lda % 4 == 0 && g.K % 4 == 0)) return false;
    if (bkc && !(al(g.B) && g.ldb % 4 == 0 && g.K % 4 == 0)) return false;
    const int64_t a_bytes = ((akc ? g.M : g.K) * g.lda) * 4, b_bytes = ((bkc ? g.N : g.K) * g.ldb) * 4;
    return a_bytes < (int64_t(1) << 31) && b_bytes < (int64_t(1) << 31);
}

// Backward of the FIRST Dense layer (no dX) with the whole optimizer step folded in (single-GPU training step):
// blocks [0, n_dw): dW0 = X^T dZ0 tiles + db0, Adam applied to W0 / b0 in the epilogue; blocks >= n_dw: Adam over
// the flat range that holds every other layer's parameters (their gradients were finished by earlier launches).
// The step then has no optimizer launch at all.
template <int WAVES, bool WIDE = false>
__global__ __launch_bounds__(WAVES * 64) void dense_bwd0_adam_kernel(GemmArgs gw, float* __restrict__ db, AdamEpi ad,
                                                                     int n_dw) {
    __shared__ float red[WAVES][WIDE ? 8 : 4][64];
    __shared__ float bsum[WAVES][WIDE ? 2 : 1][64];
    if ((int)blockIdx.x < n_dw) {
        if constexpr (WIDE) dw_tile_wide<WAVES, true, false>(gw, db, (int)blockIdx.x, red, bsum, &ad);
        else small_tile<false, false, WAVES, true>(gw, db, (int)blockIdx.x, red, reinterpret_cast<float(*)[64]>(bsum), &ad);
        return;
    }
    const float ic1 = (float)(1.0 / (1.0 - ad.pows[0])), ic2 = (float)(1.0 / (1.0 - ad.pows[1]));
    const int64_t nth = (int64_t)(gridDim.x - n_dw) * blockDim.x;
    for (int64_t i = (int64_t)(blockIdx.x - n_dw) * blockDim.x + threadIdx.x; i < ad.fn; i += nth) {
        float m = ad.fm[i], v = ad.fv[i];
        ad.fp[i] = adam_apply(ad, ic1, ic2, ad.fg[i], m, v, ad.fp[i]);
        ad.fm[i] = m;
        ad.fv[i] = v;
    }
}

// Backward of the FIRST Dense layer + gradient all-reduce + Adam in ONE launch (data-parallel step on the xGMI peer-to-peer
// transport; examples/mnist/run.py:82-83 is where the exchange sits, core/optimizer.py:67-79 the update):
//   blocks [0, n_dw): dW0 = X^T dZ0 tiles + db0 — the finished tile never goes to the gradient arena: its 64 float4 are
//     pushed straight into the recv slots of the ranks that own them (stage A of the all-reduce, tnn_p2p.hip);
//   blocks >= n_dw (always the LAST of the grid, so every tile block of this rank has been placed before one of them can
//     start to wait): stage A for the rest of the arena (the other layers' gradients and the loss, finished by earlier
//     launches), then stages B and C with Adam — tnn::p2p::allreduce_body, the body of p2p_allreduce_kernel.
// One launch and one L2 round trip less than tnn_dense_bwd + tnn_allreduce_adam, and the pollers' start-up and the
// rest-of-arena sends hide behind the product.  Slot tags: polling workgroup b advances ar_epoch[b] as in
// p2p_allreduce_kernel (the entries are equal between launches); tile block j reads ITS tag from entry j % P and then
// arrives at gate j % P, and polling workgroup b advances its entry only after all of them have — no launch-wide ticket
// (912 agent-scope atomics on one word cost 25 us, measured; here 6 or 7 per word).
struct ArTileArgs {
    float* buf;                      // gradient arena (+ slots behind it): [0, n) is reduced
    int64_t n, slice;
    int64_t w_off, b_off;            // where dW0 [M][N] and db0 [N] live in it (multiples of 4)
    int n_dw;
};

#ifdef TNN_AR_TRACE
__device__ unsigned long long g_ar_trace[1024 * 4];
#endif

template <int WAVES, bool WIDE = false>
__global__ __launch_bounds__(WAVES * 64) void dense_bwd0_allreduce_adam_kernel(GemmArgs gw, ArTileArgs f,
                                                                               tnn::p2p::LaunchCtx ctx,
                                                                               tnn::p2p::AdamTail t) {
    using namespace tnn::p2p;
    constexpr int TW = WIDE ? 32 : 16;                     // columns of a tile
    __shared__ float red[WAVES][WIDE ? 8 : 4][64];
    __shared__ float bsum[WAVES][WIDE ? 2 : 1][64];
    __shared__ __attribute__((aligned(16))) float tile[16 * TW + TW];
    const Peers& p = ctx.peers;
    const int tid = threadIdx.x, P = ctx.ar_grid;
#ifdef TNN_AR_TRACE
    if (tid == 0 && blockIdx.x < 1024) g_ar_trace[blockIdx.x * 4] = wall_clock64();
#endif
    if ((int)blockIdx.x < f.n_dw) {
        const int b = (int)blockIdx.x % P;
        const uint32_t tag = ctx.ar_epoch[b] + 1;
        // (requested in front of the product: behind it, this agent-scope load was a round trip on the way to the sends — 0.9 us
        // between "product done" and "sends issued" in profiles/r06_dp_step_stamps.txt; the word is sticky, an earlier look is as good)
        const int dead_at_entry = __hip_atomic_load(ctx.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (WIDE) dw_tile_wide<WAVES, false, true>(gw, nullptr, (int)blockIdx.x, red, bsum, nullptr, tile);
        else small_tile<false, false, WAVES, false, true>(gw, nullptr, (int)blockIdx.x, red, reinterpret_cast<float(*)[64]>(bsum), nullptr, tile);
        __syncthreads();
#ifdef TNN_AR_TRACE
        if (tid == 0 && blockIdx.x < 1024) g_ar_trace[blockIdx.x * 4 + 1] = wall_clock64();
#endif
        constexpr int Q = TW / 4;                              // float4 per tile row
        if (tid < 16 * Q + Q && dead_at_entry == 0) {
            int tm, tn;
            small_tile_coords(gw, (int)blockIdx.x, tm, tn);
            const int64_t m0 = (int64_t)tm * 16, n0 = (int64_t)tn * TW;
            int64_t e;
            bool live;
            f32x4 v;
            if (tid < 16 * Q) {
                const int row = tid / Q, c4 = tid % Q;
                live = m0 + row < gw.M && n0 + 4 * c4 < gw.N;
                e = f.w_off + (m0 + row) * gw.ldc + n0 + 4 * c4;
                v = *reinterpret_cast<const f32x4*>(tile + row * TW + 4 * c4);
            } else {
                const int c4 = tid - 16 * Q;
                live = tm == 0 && n0 + 4 * c4 < gw.N;
                e = f.b_off + n0 + 4 * c4;
                v = *reinterpret_cast<const f32x4*>(tile + 16 * TW + 4 * c4);
            }
            if (live) {
                // (the arena of a latency-size net is far below 2^31 elements: 32-bit division, a 64-bit one costs ~10x the instructions)
                const int q = f.n < (int64_t(1) << 31) ? (int)((uint32_t)e / (uint32_t)f.slice) : (int)(e / f.slice);
                ll_send(p.base[q] + ll_recv_off(p, p.rank, (e - (int64_t)q * f.slice) / 4), v, tag);
            }
        }
        __syncthreads();                                  // every thread has its tag, the stores are issued
        if (tid == 0) __hip_atomic_fetch_add(ctx.ar_gate + b * GATE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TNN_AR_TRACE
        if (tid == 0 && blockIdx.x < 1024) g_ar_trace[blockIdx.x * 4 + 2] = wall_clock64();
#endif
        return;
    }
    const int b = (int)blockIdx.x - f.n_dw;               // polling workgroup b of P
    const uint32_t tag = ctx.ar_epoch[b] + 1;
    unsigned* const gate = ctx.ar_gate + b * GATE_STRIDE;
    SkipRanges skip;
    skip.lo0 = f.w_off; skip.hi0 = f.w_off + gw.M * gw.N;
    skip.lo1 = f.b_off; skip.hi1 = f.b_off + gw.N;
#ifdef TNN_AR_TRACE
    if (blockIdx.x < 1024) skip.trace = g_ar_trace + blockIdx.x * 4;
#endif
    allreduce_body<true, 2>(p, f.buf, f.n, f.slice, tag, ctx.dead, ctx.timeout_ticks, t, (int64_t)b * (WAVES * 64) + tid,
                         (int64_t)P * (WAVES * 64), skip, gate, b < f.n_dw ? (unsigned)((f.n_dw - b + P - 1) / P) : 0u);
    __syncthreads();                                      // every thread has read epoch[b]
    if (tid == 0) {
        __hip_atomic_store(gate, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // its producers have all arrived
        ctx.ar_epoch[b] = tag;
    }
}

template <int WAVES>
int launch_small(GemmArgs& g, int transA, int transB, float* colsum) {
    g.tiles_m = (int)((g.M + 15) / 16);
    g.tiles_n = (int)((g.N + 15) / 16);
    g.splits = 1;
    g.ws = nullptr;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
    hipStream_t s = tnn::stream();
    const bool fast = small_fast_ok(g, transA, transB);
    pick_xcd_cut(g);
#define TNN_SMALL(AKC, BKC)                                                                                        \
    do {                                                                                                           \
        if (fast) hipLaunchKernelGGL((gemm_small_f32_kernel<AKC, BKC, WAVES, true>), grid, WAVES * 64, 0, s, g, colsum); \
        else hipLaunchKernelGGL((gemm_small_f32_kernel<AKC, BKC, WAVES, false>), grid, WAVES * 64, 0, s, g, colsum);    \
    } while (0)
    if (!transA && !transB) TNN_SMALL(true, false);
    else if (!transA && transB) TNN_SMALL(true, true);
    else if (transA && !transB) TNN_SMALL(false, false);
    else TNN_SMALL(false, true);
#undef TNN_SMALL
    TNN_LAUNCH_OK();
    return 0;
}

bool use_small_path(const GemmArgs& g) {
    if (const char* e = getenv("TNN_GEMM_SMALL")) return atoi(e) != 0;
    if (getenv("TNN_GEMM_CFG")) return false;
    double flop = 2.0 * (double)g.M * (double)g.N * (double)g.K;
    int64_t tiles = ((g.M + 15) / 16) * ((g.N + 15) / 16);
    // measured on the MNIST net's layers (whole step, one GPU): bs 256 36 us; bs 512 70 us with the tiled kernel (205 MFLOP in 32
    // tiles of 64 x 64 + split-K + its reduce launch) against 57 us here; bs 1024 77 / 76 us — the switch-over sits at the
    // largest product of the bs-1024 step (2 x 1024 x 784 x 256 = 411 MFLOP)
    return flop <= 4.2e8 && tiles <= 8192;
}

int gemm_small(GemmArgs& g, int transA, int transB, float* colsum) {
    int nchunks = (int)((g.K + 15) / 16);
    // (round 6, fwd0 of the MNIST net — 49 chunks — with 8 waves and all 7 chunks of a wave requested in one round instead of 16
    // waves: 21.68 against 21.33 us per step, three alternating process pairs on one box; the 16-wave workgroups' slower placement
    // — 1.5 us from the first to the last workgroup entry, profiles/r06_stepA_stamps.txt — costs less than the longer MFMA chain)
    if (nchunks <= 16) return launch_small<4>(g, transA, transB, colsum);
    if (nchunks <= 48) return launch_small<8>(g, transA, transB, colsum);
    return launch_small<16>(g, transA, transB, colsum);
}

// ------------------------------------------------------------------------------ mid-size latency GEMM
// Between the 16 x 16 latency kernel above (MNIST layers at 128-256 rows) and the LDS-tiled kernel (>= 256 tiles of 128 x 64):
// the bs-512 / bs-1024 MNIST step has products of 200-400 MFLOP (1024 x 256 x 784) that are 1024 tiles x 16 waves for the
// former (14.8 us) and 32-64 tiles + split-K + a reduce launch for the latter.  Same idea as small_tile_fast with a
// 32 x 32 tile on v_mfma_f32_32x32x2_f32: one workgroup of 8 waves per output tile, the waves split K in 8-deep chunks
// (4 MFMAs), fragments come straight from global memory through buffer loads (out-of-range -> 0, no exec-mask branches),
// up to 4 chunks of loads in flight per wave, partial tiles meet in LDS, coalesced epilogue.  Within an 8-deep chunk the
// lane group g = lane / 32 owns k = 4 g .. 4 g + 3 and MFMA s contracts element s of both groups — a permutation of the
// contraction index that lets K-contiguous operands arrive as ONE 16-B load per lane and chunk.
// One 32 x 32 tile of the mid-size kernel (workgroup `block` of the tile grid).  TO_LDS: the finished tile goes to out_lds [32][32]
// (row-major; entries outside the matrix are not written) and the column sums of the first tile row to out_lds[1024 .. 1056)
// instead of memory — for a caller that sends them elsewhere itself (dense_bwd0_mid_allreduce_adam_kernel).
template <bool AKC, bool BKC, bool ADAM, bool TO_LDS = false>
__device__ __forceinline__ void mid_tile(const GemmArgs& g, float* __restrict__ colsum, const AdamEpi& ad, const int block,
                                         float (*red)[16][64], float (*bsum)[64], float* out_lds = nullptr) {
    constexpr int WAVES = 8, MAXC = 4;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    constexpr uint32_t OOB = 0xffffffffu;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    int tm, tn;
    if (g.xg_m) {
        const int xcd = block & 7, idx = block >> 3, pm = g.tiles_m / g.xg_m, pn = g.tiles_n / g.xg_n;
        tm = (xcd % g.xg_m) * pm + idx % pm;
        tn = (xcd / g.xg_m) * pn + idx / pm;
    } else {
        tm = block % g.tiles_m;
        tn = block / g.tiles_m;
    }
    const int64_t m0 = (int64_t)tm * 32, n0 = (int64_t)tn * 32;
    const int64_t am = m0 + l31, bn = n0 + l31;
    const bool a_ok = am < g.M, b_ok = bn < g.N;
    const uint32_t K = (uint32_t)g.K, lda4 = (uint32_t)g.lda * 4u, ldb4 = (uint32_t)g.ldb * 4u;
    const int nch = (int)((g.K + 7) / 8);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A), 0, (uint32_t)(((AKC ? g.M : g.K) - 1) * g.lda + (AKC ? g.K : g.M)) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.B), 0, (uint32_t)(((BKC ? g.N : g.K) - 1) * g.ldb + (BKC ? g.K : g.N)) * 4u, 0x00020000);
    const uint32_t a_lane = AKC ? (uint32_t)am * lda4 + (uint32_t)lhi * 16u : (uint32_t)lhi * 4u * lda4 + (uint32_t)am * 4u;
    const uint32_t b_lane = BKC ? (uint32_t)bn * ldb4 + (uint32_t)lhi * 16u : (uint32_t)lhi * 4u * ldb4 + (uint32_t)bn * 4u;
    const uint32_t a_chunk = AKC ? 32u : 8u * lda4, b_chunk = BKC ? 32u : 8u * ldb4;

    // epilogue: thread (lane, r0 = tid / 64) finishes accumulator registers r0 and r0 + 8 of every lane slot
    int64_t e_row[2], e_col;
    bool e_live[2];
    float e_pre[2] = {0.f, 0.f}, a_p[2] = {0.f, 0.f}, a_m[2] = {0.f, 0.f}, a_v[2] = {0.f, 0.f};
    e_col = n0 + l31;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rr = (tid >> 6) + 8 * u;
        e_row[u] = m0 + (rr & 3) + 8 * (rr >> 2) + 4 * lhi;
        e_live[u] = e_row[u] < g.M && e_col < g.N;
        if (e_live[u]) {
            const int64_t o = e_row[u] * g.ldc + e_col;
            if (g.epi == EPI_BIAS_ACT) e_pre[u] = g.bias ? g.bias[e_col] : 0.f;
            else if (g.epi == EPI_MASK) e_pre[u] = g.Y[e_row[u] * g.ldy + e_col];
            else if (g.beta != 0.f) e_pre[u] = g.C[o];
            if constexpr (ADAM) { a_p[u] = ad.pw[o]; a_m[u] = ad.mw[o]; a_v[u] = ad.vw[o]; }
        }
    }
    float ab_p = 0.f, ab_m = 0.f, ab_v = 0.f, ic1 = 0.f, ic2 = 0.f;
    if constexpr (ADAM) {
        ic1 = (float)(1.0 / (1.0 - ad.pows[0]));
        ic2 = (float)(1.0 / (1.0 - ad.pows[1]));
        if (tm == 0 && tid < 32 && n0 + tid < g.N) { ab_p = ad.pb[n0 + tid]; ab_m = ad.mb[n0 + tid]; ab_v = ad.vb[n0 + tid]; }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bs = 0.f;
    for (int c0 = wid; c0 < nch; c0 += WAVES * MAXC) {
        float a[MAXC][4], b[MAXC][4];
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            const uint32_t c = (uint32_t)(c0 + u * WAVES);
            const uint32_t k = c * 8u + (uint32_t)lhi * 4u;
            if ((int)c >= nch) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[u][j] = 0.f; b[u][j] = 0.f; }
                continue;
            }
            if constexpr (AKC) {
                const uint32_t off = (a_ok && k < K) ? a_lane + c * a_chunk : OOB;            // K % 4 == 0: all in or all out
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[u][j] = __uint_as_float(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t off = (a_ok && k + j < K) ? a_lane + c * a_chunk + (uint32_t)j * lda4 : OOB;
                    a[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off, 0, 0));
                }
            }
            if constexpr (BKC) {
                const uint32_t off = (b_ok && k < K) ? b_lane + c * b_chunk : OOB;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, off, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[u][j] = __uint_as_float(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t off = (b_ok && k + j < K) ? b_lane + c * b_chunk + (uint32_t)j * ldb4 : OOB;
                    b[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rsrc, off, 0, 0));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], b[u][j], acc, 0, 0, 0);
            bs += (b[u][0] + b[u][1]) + (b[u][2] + b[u][3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][r][lane] = acc[r];
    bsum[wid][lane] = bs;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rr = (tid >> 6) + 8 * u;
        float sres = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sres += red[w][rr][lane];
        if (e_live[u]) {
            const float val = finish_epilogue(g, sres, e_pre[u]);
            const int64_t o = e_row[u] * g.ldc + e_col;
            if constexpr (TO_LDS) out_lds[(int)(e_row[u] - m0) * 32 + l31] = val;
            else if (!ADAM || g.C != nullptr) g.C[o] = val;
            if constexpr (ADAM) {
                ad.pw[o] = adam_apply(ad, ic1, ic2, val, a_m[u], a_v[u], a_p[u]);
                ad.mw[o] = a_m[u];
                ad.vw[o] = a_v[u];
            }
        }
    }
    if ((TO_LDS || colsum != nullptr) && tm == 0 && tid < 32 && n0 + tid < g.N) {
        float sres = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sres += bsum[w][tid] + bsum[w][32 + tid];
        if constexpr (TO_LDS) out_lds[1024 + tid] = sres;
        else colsum[n0 + tid] = sres;
        if constexpr (ADAM) {
            ad.pb[n0 + tid] = adam_apply(ad, ic1, ic2, sres, ab_m, ab_v, ab_p);
            ad.mb[n0 + tid] = ab_m;
            ad.vb[n0 + tid] = ab_v;
        }
    }
}


template <bool AKC, bool BKC, bool ADAM>
__global__ __launch_bounds__(512) void gemm_mid_f32_kernel(GemmArgs g, float* __restrict__ colsum, AdamEpi ad, int n_tiles) {
    constexpr int WAVES = 8, MAXC = 4;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    constexpr uint32_t OOB = 0xffffffffu;
    if constexpr (ADAM) {
        if ((int)blockIdx.x >= n_tiles) {               // trailing workgroups: Adam over the other layers' flat range
            const float ic1 = (float)(1.0 / (1.0 - ad.pows[0])), ic2 = (float)(1.0 / (1.0 - ad.pows[1]));
            const int64_t nth = (int64_t)(gridDim.x - n_tiles) * blockDim.x;
            for (int64_t i = (int64_t)(blockIdx.x - n_tiles) * blockDim.x + threadIdx.x; i < ad.fn; i += nth) {
                float m = ad.fm[i], v = ad.fv[i];
                ad.fp[i] = adam_apply(ad, ic1, ic2, ad.fg[i], m, v, ad.fp[i]);
                ad.fm[i] = m;
                ad.fv[i] = v;
            }
            return;
        }
    }
    __shared__ float red[WAVES][16][64];
    __shared__ float bsum[WAVES][64];
    mid_tile<AKC, BKC, ADAM>(g, colsum, ad, (int)blockIdx.x, red, bsum);
}

// the mid-size kernel takes 16-B loads on its K-contiguous operands: alignment, ld % 4 == 0, K % 4 == 0, extents below 2 GiB
bool mid_ok(const GemmArgs& g, int transA, int transB) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool akc = !transA, bkc = transB != 0;
    if (akc && !(al(g.A) && g.lda % 4 == 0 && g.K % 4 == 0)) return false;
    if (bkc && !(al(g.B) && g.ldb % 4 == 0 && g.K % 4 == 0)) return false;
    const int64_t a_bytes = ((akc ? g.M : g.K) * g.lda) * 4, b_bytes = ((bkc ? g.N : g.K) * g.ldb) * 4;
    return a_bytes < (int64_t(1) << 31) && b_bytes < (int64_t(1) << 31);
}

// products the 16 x 16 kernel would take but that are big enough for 32 x 32 tiles to pay: >= 150 MFLOP and >= 192 tiles
// (measured on the MNIST step: 1024 x 256 x 784 forward 14.8 -> 8.5 us in 256 tiles, 784 x 256 x 1024 / x 512 weight gradient
// with Adam 17.4 -> 10.9 / 10.4 -> 7.2 us in 200 tiles; the 512-row forward in 128 tiles was no faster: 8.4 vs 8.0 us)
bool use_mid_path(const GemmArgs& g, int transA, int transB) {
    const double flop = 2.0 * (double)g.M * (double)g.N * (double)g.K;
    const int64_t tiles = ((g.M + 31) / 32) * ((g.N + 31) / 32);
    return flop >= 1.5e8 && tiles >= 192 && mid_ok(g, transA, transB);
}

void mid_geometry(GemmArgs& g) {
    g.tiles_m = (int)((g.M + 31) / 32);
    g.tiles_n = (int)((g.N + 31) / 32);
    g.splits = 1;
    g.ws = nullptr;
    pick_xcd_cut(g);
}

int gemm_mid(GemmArgs& g, int transA, int transB, float* colsum) {
    mid_geometry(g);
    const int tiles = g.tiles_m * g.tiles_n;
    hipStream_t s = tnn::stream();
    AdamEpi none = {};
#define TNN_MID(AKC, BKC) hipLaunchKernelGGL((gemm_mid_f32_kernel<AKC, BKC, false>), tiles, 512, 0, s, g, colsum, none, tiles)
    if (!transA && !transB) TNN_MID(true, false);
    else if (!transA && transB) TNN_MID(true, true);
    else if (transA && !transB) TNN_MID(false, false);
    else TNN_MID(false, true);
#undef TNN_MID
    TNN_LAUNCH_OK();
    return 0;
}

// dense_bwd0_allreduce_adam_kernel for shards of more than 256 rows (the 32 x 32 tile form: 200 tiles of the 784 x 256 weight
// gradient at 512 / 1024 rows per rank): blocks [0, n_dw) compute a tile and push its 256 float4 (+ 8 of the column sums in the
// first tile row) straight into the owners' receive slots; blocks >= n_dw are the transport's polling workgroups (stage A for the
// rest of the arena, stages B and C with Adam) — one launch instead of tnn_dense_bwd + tnn_allreduce_adam, and the pollers'
// start-up and rest-of-arena sends hide behind the product (round 6; tags and gates as in the 16 x 16 form).
__global__ __launch_bounds__(512) void dense_bwd0_mid_allreduce_adam_kernel(GemmArgs gw, ArTileArgs f, tnn::p2p::LaunchCtx ctx,
                                                                            tnn::p2p::AdamTail t) {
    using namespace tnn::p2p;
    __shared__ float red[8][16][64];
    __shared__ float bsum[8][64];
    __shared__ __attribute__((aligned(16))) float tile[32 * 32 + 32];
    const Peers& p = ctx.peers;
    const int tid = threadIdx.x, P = ctx.ar_grid;
    if ((int)blockIdx.x < f.n_dw) {
        const int b = (int)blockIdx.x % P;
        const uint32_t tag = ctx.ar_epoch[b] + 1;
        const AdamEpi none = {};
        const int dead_at_entry = __hip_atomic_load(ctx.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (see the 16-row kernel)
        mid_tile<false, false, false, true>(gw, nullptr, none, (int)blockIdx.x, red, bsum, tile);
        __syncthreads();
        if (tid < 264 && dead_at_entry == 0) {
            int tm, tn;
            small_tile_coords(gw, (int)blockIdx.x, tm, tn);            // (the same XCD-aware order as mid_tile)
            const int64_t m0 = (int64_t)tm * 32, n0 = (int64_t)tn * 32;
            int64_t e;
            bool live;
            f32x4 v;
            if (tid < 256) {
                const int row = tid >> 3, c4 = tid & 7;
                live = m0 + row < gw.M && n0 + 4 * c4 < gw.N;
                e = f.w_off + (m0 + row) * gw.ldc + n0 + 4 * c4;
                v = *reinterpret_cast<const f32x4*>(tile + row * 32 + 4 * c4);
            } else {
                const int c4 = tid - 256;
                live = tm == 0 && n0 + 4 * c4 < gw.N;
                e = f.b_off + n0 + 4 * c4;
                v = *reinterpret_cast<const f32x4*>(tile + 1024 + 4 * c4);
            }
            if (live) {
                // (the arena of a latency-size net is far below 2^31 elements: 32-bit division, a 64-bit one costs ~10x the instructions)
                const int q = f.n < (int64_t(1) << 31) ? (int)((uint32_t)e / (uint32_t)f.slice) : (int)(e / f.slice);
                ll_send(p.base[q] + ll_recv_off(p, p.rank, (e - (int64_t)q * f.slice) / 4), v, tag);
            }
        }
        __syncthreads();                                  // every thread has its tag, the stores are issued
        if (tid == 0) __hip_atomic_fetch_add(ctx.ar_gate + b * GATE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int b = (int)blockIdx.x - f.n_dw;               // polling workgroup b of P
    const uint32_t tag = ctx.ar_epoch[b] + 1;
    unsigned* const gate = ctx.ar_gate + b * GATE_STRIDE;
    SkipRanges skip;
    skip.lo0 = f.w_off; skip.hi0 = f.w_off + gw.M * gw.N;
    skip.lo1 = f.b_off; skip.hi1 = f.b_off + gw.N;
    allreduce_body<true, 2>(p, f.buf, f.n, f.slice, tag, ctx.dead, ctx.timeout_ticks, t, (int64_t)b * 512 + tid, (int64_t)P * 512, skip,
                            gate, b < f.n_dw ? (unsigned)((f.n_dw - b + P - 1) / P) : 0u);
    __syncthreads();                                      // every thread has read epoch[b]
    if (tid == 0) {
        __hip_atomic_store(gate, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // its producers have all arrived
        ctx.ar_epoch[b] = tag;
    }
}

// ------------------------------------------------------------------------------ f64 (exact mode)
// Plain LDS-tiled VALU kernel, 64x64 tile, 4x4 micro-tile per thread; not on the measured path.
struct GemmArgsD {
    const double* A;
    const double* B;
    double* C;
    int64_t M, N, K, sam, sak, sbk, sbn, ldc;
    double alpha, beta;
    int epi;
    const double* bias;
    int act, relu_sign;
    const double* Y;
    int64_t ldy;
};

__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgsD g) {
    __shared__ double As[16][64 + 1], Bs[16][64 + 1];
    int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    double acc[4][4] = {};
    for (int64_t k0 = 0; k0 < g.K; k0 += 16) {
        for (int f = threadIdx.x; f < 16 * 64; f += 256) {
            int kk = f / 64, mm = f % 64;
            int64_t gm = m0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < g.M && gk < g.K) ? g.A[gm * g.sam + gk * g.sak] : 0.0;
            int64_t gn = n0 + mm;
            Bs[kk][mm] = (gn < g.N && gk < g.K) ? g.B[gk * g.sbk + gn * g.sbn] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t row = m0 + ty * 4 + i, col = n0 + tx * 4 + j;
            if (row >= g.M || col >= g.N) continue;
            double v = acc[i][j];
            if (g.epi == EPI_AXPBY) {
                v = g.alpha * v;
                if (g.beta != 0.0) v += g.beta * g.C[row * g.ldc + col];
            } else if (g.epi == EPI_BIAS_ACT) {
                v += g.bias ? g.bias[col] : 0.0;
                if (g.act == TNN_ACT_RELU) {
                    if (g.relu_sign) v = v < 0.0 ? -0.0 : fabs(v);
                    else v = v < 0.0 ? 0.0 : v;
                }
            } else {
                v = signbit(g.Y[row * g.ldy + col]) ? 0.0 : v;
            }
            g.C[row * g.ldc + col] = v;
        }
}

// ------------------------------------------------------------------------------ host dispatch
template <int BM, int BN, int BK, int WM, int WN>
int launch_cfg(GemmArgs& g, int transA, int transB, int splits) {
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (int)((g.N + BN - 1) / BN);
    int64_t ktiles = (g.K + BK - 1) / BK;
    if (splits > ktiles) splits = (int)ktiles;
    if (splits < 1) splits = 1;
    int64_t tiles_per_split = (ktiles + splits - 1) / splits;
    if (tiles_per_split < 1) tiles_per_split = 1;   // K == 0: one empty pass, C = beta*C / bias
    splits = ktiles > 0 ? (int)((ktiles + tiles_per_split - 1) / tiles_per_split) : 1;
    g.k_per_split = tiles_per_split * BK;
    g.splits = splits;
    g.ws = nullptr;
    g.group_m = 8;
    if (const char* e = getenv("TNN_GEMM_GROUP_M")) g.group_m = atoi(e) > 0 ? atoi(e) : 8;       // tuning override (sweeps)
#ifdef TNN_GEMM_TRACE
    g.trace = nullptr;
    if (const char* e = getenv("TNN_GEMM_TRACE_PTR"))
        if (splits == 1) g.trace = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    void* ws = nullptr;
    if (splits > 1) {
        if (tnn_malloc((size_t)splits * g.M * g.N * sizeof(float), &ws)) return 1;
        g.ws = (float*)ws;
    }
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)splits);
    constexpr int NT = WM * WN * 64;
    hipStream_t s = tnn::stream();
    const bool vec = g.vecA && g.vecB;
#define TNN_LAUNCH_GEMM(AKC, BKC)                                                                          \
    do {                                                                                                   \
        if (vec)                                                                                           \
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, WM, WN, AKC, BKC, true>), grid, NT, 0, s, g); \
        else                                                                                               \
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, WM, WN, AKC, BKC, false>), grid, NT, 0, s, g); \
    } while (0)
    if (!transA && !transB) TNN_LAUNCH_GEMM(true, false);
    else if (!transA && transB) TNN_LAUNCH_GEMM(true, true);
    else if (transA && !transB) TNN_LAUNCH_GEMM(false, false);
    else TNN_LAUNCH_GEMM(false, true);
#undef TNN_LAUNCH_GEMM
    if (splits > 1) {
        hipLaunchKernelGGL(splitk_reduce_kernel, tnn::stream_grid(g.M * g.N, 256), 256, 0, s, g);
        tnn_free(ws);
    }
    TNN_LAUNCH_OK();
    return 0;
}

int gemm_f32(GemmArgs& g, int transA, int transB, float* colsum = nullptr) {
    // 16-B loads need: contiguous extent and leading dimension multiples of 4, base 16-B aligned
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // ... and, for the 32-bit per-thread offsets of the fast loader, an operand below 4 GiB
    const int64_t a_bytes = (transA ? g.K : g.M) * g.lda * 4, b_bytes = (transB ? g.N : g.K) * g.ldb * 4;
    g.vecA = al(g.A) && g.lda % 4 == 0 && ((transA ? g.M : g.K) % 4 == 0) && a_bytes < (int64_t(1) << 32);
    g.vecB = al(g.B) && g.ldb % 4 == 0 && ((transB ? g.K : g.N) % 4 == 0) && b_bytes < (int64_t(1) << 32);
    if (use_small_path(g)) return use_mid_path(g, transA, transB) ? gemm_mid(g, transA, transB, colsum)
                                                                    : gemm_small(g, transA, transB, colsum);
    if (colsum != nullptr) {   // large shapes: the column sum is a separate (HBM-bound, <1 % of the time) pass
        if (int rc = tnn_reduce(TNN_RSUM, g.B, colsum, 1, g.K, g.N, TNN_F32)) return rc;
    }

    const int cus = tnn::num_cus();
    g.staged_c = 1;
    int cfg = -1, splits = 0;
    if (const char* e = getenv("TNN_GEMM_CFG")) cfg = atoi(e);       // tuning override
    if (const char* e = getenv("TNN_GEMM_SPLITK")) splits = atoi(e);
    // Measured on MI355X (tools/gemm_sweep.py, 512x4096x4096 NN/NT and 4096x4096x512 TN): the 128x64 tile
    // (4 waves of 64x32, 52 KB LDS, 3 workgroups per CU) is the fastest of the four on every layout
    // (100-105 TFLOP/s); 64x64 only wins when the problem has too few 128x64 tiles to occupy the chip.
    int64_t t128x64 = ((g.M + 127) / 128) * ((g.N + 63) / 64);
    int64_t t64 = ((g.M + 63) / 64) * ((g.N + 63) / 64);
    if (cfg < 0) cfg = (t128x64 >= cus / 2) ? 3 : 2;
    // Round 3, same-box A/B (tools/gemm_sweep.py, three boxes): for the 512-row products of config C (NN / NT, one tile per
    // CU either way) the 64 x 128 tile is 140-143 us on every box where 128 x 64 moves between 139 and 157 — never slower, up
    // to 9 % faster.  (The TN weight-gradient shape shows no such preference: 128 x 64 stays.)
    const int64_t t64x128 = ((g.M + 63) / 64) * ((g.N + 127) / 128);
    if (cfg == 3 && !getenv("TNN_GEMM_CFG") && !transA && g.M <= 1024 && t64x128 >= cus / 2 && t64x128 <= 2 * cus) cfg = 1;
    // (Round 3, measured and not kept: the same 128 x 64 / 64 x 128 tiles with EIGHT waves of 32 x 32 — two per SIMD, so one
    // wave's MFMAs can cover the other's barrier and wait states.  Plain products on one box: TN 156 us against 162 (64 x 128)
    // and 172 (128 x 64), NN 146 against 148, NT 144-147 against 143; inside config C's step, where the TN launches carry the
    // Adam epilogue, 0.739-0.745 ms against 0.736 for the four-wave 128 x 64 tile.)
    if (splits <= 0) {
        splits = 1;
        int64_t tiles = cfg == 3 ? t128x64 : cfg == 2 ? t64 : cfg == 0 ? ((g.M + 127) / 128) * ((g.N + 127) / 128)
                                                              : ((g.M + 63) / 64) * ((g.N + 127) / 128);
        if (tiles < cus) {
            // fewer tiles than CUs: split K until every CU has a workgroup, keeping >= 8 K-tiles (256 deep)
            // per split so that the extra reduce pass (M*N*splits*4 B of traffic) stays small
            splits = (int)((cus + tiles - 1) / tiles);
            int64_t max_splits = g.K / 256;
            if (splits > max_splits) splits = (int)max_splits;
            if (splits < 1) splits = 1;
            if (splits > 16) splits = 16;
        }
    }
    switch (cfg) {
        case 0: return launch_cfg<128, 128, 32, 2, 2>(g, transA, transB, splits);
        case 1: return launch_cfg<64, 128, 32, 2, 2>(g, transA, transB, splits);
        case 2: return launch_cfg<64, 64, 32, 2, 2>(g, transA, transB, splits);
        case 3: return launch_cfg<128, 64, 32, 2, 2>(g, transA, transB, splits);
    }
    tnn::set_error("tnn_gemm: unknown tile configuration %d", cfg);
    return 2;
}

int gemm_f64(int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
             const void* B, int64_t ldb, void* C, int64_t ldc, double alpha, double beta, int epi,
             const void* bias, int act, int relu_sign, const void* Y, int64_t ldy) {
    GemmArgsD g;
    g.relu_sign = relu_sign;
    g.A = (const double*)A; g.B = (const double*)B; g.C = (double*)C;
    g.M = M; g.N = N; g.K = K;
    g.sam = transA ? 1 : lda; g.sak = transA ? lda : 1;
    g.sbk = transB ? 1 : ldb; g.sbn = transB ? ldb : 1;
    g.ldc = ldc; g.alpha = alpha; g.beta = beta; g.epi = epi;
    g.bias = (const double*)bias; g.act = act; g.Y = (const double*)Y; g.ldy = ldy;
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    hipLaunchKernelGGL(gemm_f64_kernel, grid, 256, 0, tnn::stream(), g);
    TNN_LAUNCH_OK();
    return 0;
}

int check_shapes(const char* fn, int transA, int transB, int64_t M, int64_t N, int64_t K,
                 int64_t lda, int64_t ldb, int64_t ldc) {
    TNN_REQUIRE(M >= 0 && N >= 0 && K >= 0, "%s: negative extent", fn);
    TNN_REQUIRE(lda >= (transA ? M : K), "%s: lda %lld too small", fn, (long long)lda);
    TNN_REQUIRE(ldb >= (transB ? K : N), "%s: ldb %lld too small", fn, (long long)ldb);
    TNN_REQUIRE(ldc >= N, "%s: ldc %lld too small", fn, (long long)ldc);
    TNN_REQUIRE(M < (1LL << 31) && N < (1LL << 31), "%s: extent exceeds 2^31", fn);
    return 0;
}

// fallback of tnn_dense_fwd_head_partials for shapes the latency kernel does not take: the same partial sums from the
// finished output, one thread per (tile, row, class)
__global__ __launch_bounds__(256) void head_partials_kernel(const float* __restrict__ out, int64_t M, int64_t N, int64_t ldc,
                                                            const float* __restrict__ hw, int hc, float* __restrict__ hz) {
    const int64_t tiles_n = (N + 15) / 16, total = tiles_n * M * hc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % hc);
        const int64_t row = (i / hc) % M, tn = i / (hc * M);
        float acc = 0.f;
        for (int col = 0; col < 16 && tn * 16 + col < N; ++col)
            acc = fmaf(out[row * ldc + tn * 16 + col], hw[(tn * 16 + col) * hc + c], acc);
        hz[i] = acc;
    }
}

}  // namespace

extern "C" {

int tnn_gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha, const void* A,
             int64_t lda, const void* B, int64_t ldb, double beta, void* C, int64_t ldc, int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_gemm", transA, transB, M, N, K, lda, ldb, ldc)) return rc;
    if (M == 0 || N == 0) return 0;
    if (dtype == TNN_F64)
        return gemm_f64(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, EPI_AXPBY,
                        nullptr, 0, 0, nullptr, 0);
    TNN_REQUIRE(dtype == TNN_F32, "tnn_gemm: dtype %d is not a float type", dtype);
    GemmArgs g = {};
    g.A = (const float*)A; g.B = (const float*)B; g.C = (float*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = (float)alpha; g.beta = (float)beta; g.epi = EPI_AXPBY;
    return gemm_f32(g, transA, transB);
}

int tnn_gemm_tn_colsum(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* G,
                       int64_t ldg, void* dW, int64_t ldc, void* db, int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_gemm_tn_colsum", 1, 0, M, N, K, lda, ldg, ldc)) return rc;
    TNN_REQUIRE(ldg == N || db == nullptr, "tnn_gemm_tn_colsum: the column sum needs a dense G (ldg == N)");
    if (N == 0) return 0;
    if (dtype == TNN_F64) {
        if (M > 0)
            if (int rc = gemm_f64(1, 0, M, N, K, A, lda, G, ldg, dW, ldc, 1.0, 0.0, EPI_AXPBY, nullptr, 0, 0,
                                  nullptr, 0))
                return rc;
        return db ? tnn_reduce(TNN_RSUM, G, db, 1, K, N, TNN_F64) : 0;
    }
    TNN_REQUIRE(dtype == TNN_F32, "tnn_gemm_tn_colsum: dtype %d is not a float type", dtype);
    if (M == 0) return db ? tnn_reduce(TNN_RSUM, G, db, 1, K, N, TNN_F32) : 0;
    GemmArgs g = {};
    g.A = (const float*)A; g.B = (const float*)G; g.C = (float*)dW;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldg; g.ldc = ldc;
    g.alpha = 1.f; g.beta = 0.f; g.epi = EPI_AXPBY;
    return gemm_f32(g, 1, 0, (float*)db);
}

int tnn_gemm_tn_adam(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* G, int64_t ldg, void* g_out,
                     void* p, void* m, void* v, double lr, double b1, double b2, double eps, const void* pows_f64,
                     int dtype) {
    return tnn_gemm_tn_adam_bias(M, N, K, A, lda, G, ldg, g_out, p, m, v, nullptr, nullptr, nullptr, nullptr, lr, b1, b2, eps,
                                 pows_f64, dtype);
}

int tnn_gemm_tn_adam_bias(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* G, int64_t ldg, void* g_out,
                          void* p, void* m, void* v, void* db, void* pb, void* mb, void* vb, double lr, double b1, double b2,
                          double eps, const void* pows_f64, int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_gemm_tn_adam", 1, 0, M, N, K, lda, ldg, N)) return rc;
    TNN_REQUIRE((pb == nullptr) == (mb == nullptr) && (pb == nullptr) == (vb == nullptr) && (pb == nullptr || db != nullptr),
                "tnn_gemm_tn_adam_bias: pb / mb / vb go together and need db");
    TNN_REQUIRE(db == nullptr || ldg == N, "tnn_gemm_tn_adam_bias: the column sum needs a dense G (ldg == N)");
    TNN_REQUIRE(p && m && v && pows_f64, "tnn_gemm_tn_adam: p, m, v and pows are required");
    TNN_REQUIRE(dtype == TNN_F32 || dtype == TNN_F64, "tnn_gemm_tn_adam: dtype %d is not a float type", dtype);
    if (M == 0 || N == 0) return 0;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    GemmArgs g = {};
    g.A = (const float*)A; g.B = (const float*)G; g.C = (float*)g_out;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldg; g.ldc = N;
    g.alpha = 1.f; g.beta = 0.f; g.epi = EPI_ADAM;
    const bool tiled = dtype == TNN_F32 && !use_small_path(g) && N % 4 == 0 && al(p) && al(m) && al(v) && al(g_out) &&
                       ((M + 127) / 128) * ((N + 63) / 64) >= tnn::num_cus() / 2 && getenv("TNN_GEMM_CFG") == nullptr &&
                       getenv("TNN_GEMM_SPLITK") == nullptr;
    if (tiled) {
        g.ad_p = (float*)p; g.ad_m = (float*)m; g.ad_v = (float*)v;
        g.ad_lr = (float)lr; g.ad_b1 = (float)b1; g.ad_b2 = (float)b2; g.ad_eps = (float)eps;
        g.ad_pows = (const double*)pows_f64;
        g.ad_guard = tnn::update_guard();
        // db (+ Adam on the bias) from the same launch: the epilogue of tile row 0 (splits == 1 here)
        const bool cs = db != nullptr;
        if (cs) { g.cs_db = (float*)db; g.cs_p = (float*)pb; g.cs_m = (float*)mb; g.cs_v = (float*)vb; }
        if (int rc = gemm_f32(g, 1, 0)) return rc;   // >= 128 tiles of 128 x 64: configuration 3, no split-K
        if (db != nullptr && !cs) {
            if (int rc = tnn_reduce(TNN_RSUM, G, db, 1, K, N, dtype)) return rc;
            if (pb) return tnn_adam_ex(pb, db, mb, vb, N, lr, b1, b2, eps, const_cast<void*>(pows_f64), nullptr, dtype, 0, nullptr, nullptr);
        }
        return 0;
    }
    // any other shape / dtype: the two launches this replaces (through a scratch gradient when none is wanted)
    void* scratch = nullptr;
    void* gw = g_out;
    const size_t esz = dtype == TNN_F64 ? 8 : 4;
    if (gw == nullptr) {
        if (tnn_malloc((size_t)(M * N) * esz, &scratch)) return 1;
        gw = scratch;
    }
    int rc = tnn_gemm_tn_colsum(M, N, K, A, lda, G, ldg, gw, N, db, dtype);
    if (!rc) rc = tnn_adam_ex(p, gw, m, v, M * N, lr, b1, b2, eps, const_cast<void*>(pows_f64), nullptr, dtype, 0, nullptr, nullptr);
    if (!rc && pb) rc = tnn_adam_ex(pb, db, mb, vb, N, lr, b1, b2, eps, const_cast<void*>(pows_f64), nullptr, dtype, 0, nullptr, nullptr);
    if (scratch) tnn_free(scratch);
    return rc;
}

int tnn_dense_bwd(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz, const void* w,
                  void* dw, void* db, void* dx, const void* mask_src, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(rows > 0 && n_in > 0 && n_out > 0, "tnn_dense_bwd: empty layer");
    TNN_REQUIRE(dx == nullptr || mask_src != nullptr, "tnn_dense_bwd: dx needs mask_src");
    if (dtype == TNN_F32 && dx != nullptr) {
        auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        GemmArgs gw = {}, gx = {};
        gw.A = (const float*)x; gw.B = (const float*)dz; gw.C = (float*)dw;
        gw.M = n_in; gw.N = n_out; gw.K = rows; gw.lda = n_in; gw.ldb = n_out; gw.ldc = n_out;
        gw.alpha = 1.f; gw.beta = 0.f; gw.epi = EPI_AXPBY;
        gw.vecA = al(x) && n_in % 4 == 0; gw.vecB = al(dz) && n_out % 4 == 0;
        gx.A = (const float*)dz; gx.B = (const float*)w; gx.C = (float*)dx;
        gx.M = rows; gx.N = n_in; gx.K = n_out; gx.lda = n_out; gx.ldb = n_out; gx.ldc = n_in;
        gx.alpha = 1.f; gx.beta = 0.f; gx.epi = EPI_MASK; gx.Y = (const float*)mask_src; gx.ldy = n_in;
        gx.vecA = al(dz) && n_out % 4 == 0; gx.vecB = al(w) && n_out % 4 == 0;
        if (use_small_path(gw) && use_small_path(gx)) {
            gw.tiles_m = (int)((gw.M + 15) / 16); gw.tiles_n = (int)((gw.N + 15) / 16); gw.splits = 1;
            gx.tiles_m = (int)((gx.M + 15) / 16); gx.tiles_n = (int)((gx.N + 15) / 16); gx.splits = 1;
            int n_dw = gw.tiles_m * gw.tiles_n, n_dx = gx.tiles_m * gx.tiles_n;
            if (n_dw % 8 == 0) {                       // the dX tiles then start on XCD 0 again: both grids can be cut per XCD
                pick_xcd_cut(gw);
                pick_xcd_cut(gx);
            }
            int nchunks = (int)((std::max(gw.K, gx.K) + 15) / 16);
            hipStream_t s = tnn::stream();
            const bool fast = small_fast_ok(gw, 1, 0) && small_fast_ok(gx, 0, 1);
#define TNN_BWD_SMALL(W)                                                                                             \
    do {                                                                                                             \
        if (fast) hipLaunchKernelGGL((dense_bwd_small_kernel<W, true>), n_dw + n_dx, W * 64, 0, s, gw, (float*)db, gx, n_dw); \
        else hipLaunchKernelGGL((dense_bwd_small_kernel<W, false>), n_dw + n_dx, W * 64, 0, s, gw, (float*)db, gx, n_dw);    \
    } while (0)
            if (nchunks <= 16)
                TNN_BWD_SMALL(4);
            else if (nchunks <= 48)
                TNN_BWD_SMALL(8);
            else
                TNN_BWD_SMALL(16);
#undef TNN_BWD_SMALL
            TNN_LAUNCH_OK();
            return 0;
        }
    }
    if (int rc = tnn_gemm_tn_colsum(n_in, n_out, rows, x, n_in, dz, n_out, dw, n_out, db, dtype)) return rc;
    if (dx != nullptr)
        return tnn_gemm_mask(0, 1, rows, n_in, n_out, dz, n_out, w, n_out, mask_src, n_in, dx, n_in, dtype);
    return 0;
}

// the 16 x 32 tile form of the first layer's weight gradient (dw_tile_wide): N a multiple of 32, enough tiles to fill the chip
// either way; TNN_DW0_WIDE=0 keeps the 16 x 16 tiles (A/B measurements)
static bool dw0_wide_ok(const GemmArgs& g) {
    static const bool off = getenv("TNN_DW0_WIDE") && atoi(getenv("TNN_DW0_WIDE")) == 0;
    return !off && g.N % 32 == 0 && ((g.M + 15) / 16) * (g.N / 32) >= 256 && g.epi == EPI_AXPBY && g.beta == 0.f;
}

int tnn_dense_bwd_first_adam(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz, void* dw, void* db,
                             void* p_w, void* m_w, void* v_w, void* p_b, void* m_b, void* v_b, void* flat_p,
                             const void* flat_g, void* flat_m, void* flat_v, int64_t flat_n, double lr, double b1,
                             double b2, double eps, const void* pows_f64, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(rows > 0 && n_in > 0 && n_out > 0, "tnn_dense_bwd_first_adam: empty layer");
    TNN_REQUIRE(pows_f64 && p_w && m_w && v_w && p_b && m_b && v_b, "tnn_dense_bwd_first_adam: optimizer state is required");
    if (dtype == TNN_F32) {
        GemmArgs gw = {};
        gw.A = (const float*)x; gw.B = (const float*)dz; gw.C = (float*)dw;
        gw.M = n_in; gw.N = n_out; gw.K = rows; gw.lda = n_in; gw.ldb = n_out; gw.ldc = n_out;
        gw.alpha = 1.f; gw.beta = 0.f; gw.epi = EPI_AXPBY;
        if (use_small_path(gw)) {
            const bool mid = use_mid_path(gw, 1, 0);                 // bs >= 512: 32 x 32 tiles (gemm_mid_f32_kernel)
            if (mid) mid_geometry(gw);
            else { gw.tiles_m = (int)((gw.M + 15) / 16); gw.tiles_n = (int)((gw.N + 15) / 16); gw.splits = 1; }
            const int n_dw = gw.tiles_m * gw.tiles_n;
            if (!mid) pick_xcd_cut(gw);
            AdamEpi ad;
            ad.pw = (float*)p_w; ad.mw = (float*)m_w; ad.vw = (float*)v_w;
            ad.pb = (float*)p_b; ad.mb = (float*)m_b; ad.vb = (float*)v_b;
            ad.fp = (float*)flat_p; ad.fg = (const float*)flat_g; ad.fm = (float*)flat_m; ad.fv = (float*)flat_v;
            ad.fn = flat_n > 0 ? flat_n : 0;
            ad.lr = (float)lr; ad.b1 = (float)b1; ad.b2 = (float)b2; ad.eps = (float)eps;
            ad.pows = (const double*)pows_f64;
            const int nchunks = (int)((gw.K + 15) / 16);
            hipStream_t s = tnn::stream();
            if (mid) {
                const int extra = ad.fn > 0 ? (int)std::min<int64_t>((ad.fn + 511) / 512, 64) : 0;
                hipLaunchKernelGGL((gemm_mid_f32_kernel<false, false, true>), n_dw + extra, 512, 0, s, gw, (float*)db, ad, n_dw);
                TNN_LAUNCH_OK();
                return 0;
            }
#define TNN_BWD0(W)                                                                                         \
    do {                                                                                                    \
        const int extra = ad.fn > 0 ? (int)std::min<int64_t>((ad.fn + W * 64 - 1) / (W * 64), 64) : 0;      \
        hipLaunchKernelGGL((dense_bwd0_adam_kernel<W>), n_dw + extra, W * 64, 0, s, gw, (float*)db, ad, n_dw); \
    } while (0)
            // 16 x 32 tiles (dw_tile_wide): half the workgroups, at most two per CU for the MNIST net's 784 x 256 gradient
            if (nchunks <= 16 && dw0_wide_ok(gw)) {
                gw.tiles_n = (int)(gw.N / 32);
                pick_xcd_cut(gw);
                const int n_wide = gw.tiles_m * gw.tiles_n;
                const int extra = ad.fn > 0 ? (int)std::min<int64_t>((ad.fn + 255) / 256, 64) : 0;
                hipLaunchKernelGGL((dense_bwd0_adam_kernel<4, true>), n_wide + extra, 256, 0, s, gw, (float*)db, ad, n_wide);
                TNN_LAUNCH_OK();
                return 0;
            }
            // measured at bs 128 (8 chunks): 4 waves with two chunks each 21.5 us/step, 8 waves with one chunk each 22.2, 16
            // waves 23.7 — the launch's 784 workgroups cost more per wave than the second load round trip saves
            const int waves = nchunks <= 16 ? 4 : nchunks <= 48 ? 8 : 16;
            // (one WAVE per tile over the whole K, four tiles per workgroup — a quarter of the waves for the dispatcher to place —
            // was built and measured in round 4: bit-identical, 22.5 instead of 21.4 us/step; the per-wave chain of 64 loads
            // and 32 MFMAs costs more than the placement saves)
            if (waves == 4) TNN_BWD0(4);
            else if (waves == 8) TNN_BWD0(8);
            else TNN_BWD0(16);
#undef TNN_BWD0
            TNN_LAUNCH_OK();
            return 0;
        }
    }
    // any other shape / dtype: the launches this replaces (dw == NULL: through tnn_gemm_tn_adam_bias' scratch gradient)
    void* pows = const_cast<void*>(pows_f64);
    if (int rc = tnn_gemm_tn_adam_bias(n_in, n_out, rows, x, n_in, dz, n_out, dw, p_w, m_w, v_w, db, p_b, m_b, v_b, lr, b1, b2, eps,
                                       pows, dtype))
        return rc;
    if (flat_n > 0)
        return tnn_adam_ex(flat_p, flat_g, flat_m, flat_v, flat_n, lr, b1, b2, eps, pows, nullptr, dtype, 0, nullptr, nullptr);
    return 0;
}

#ifdef TNN_STEP_TRACE
// rows [kernel id][workgroup][4] of g_step_trace; `which` = 2 is served by tnn_head.hip's own buffer (tnn_debug_step_trace_head)
extern "C" __attribute__((visibility("default"))) int tnn_debug_step_trace(unsigned long long* out, int n) {
    TNN_CHECK_HIP(hipDeviceSynchronize());
    TNN_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_trace), (size_t)std::min(n, 4 * 1024 * 4) * 8));
    return 0;
}
#endif
#ifdef TNN_AR_TRACE
__attribute__((visibility("default"))) int tnn_debug_fh_trace(unsigned long long* out, int n) {
    TNN_CHECK_HIP(hipDeviceSynchronize());
    TNN_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fh_trace), (size_t)std::min(n, 1024) * 8));
    return 0;
}
__attribute__((visibility("default"))) int tnn_debug_ar_trace(unsigned long long* out, int n) {
    TNN_CHECK_HIP(hipDeviceSynchronize());
    TNN_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ar_trace), (size_t)std::min(n, 4096) * 8));
    return 0;
}
#endif

int tnn_dense_bwd_first_allreduce_adam(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz, void* grads,
                                       int64_t n_reduce, int64_t w_off, int64_t b_off, void* p, void* m, void* v,
                                       int64_t n_params, double lr, double b1, double b2, double eps, const void* pows_f64,
                                       int64_t scalar_index, void* scalar_dst, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(rows > 0 && n_in > 0 && n_out > 0, "tnn_dense_bwd_first_allreduce_adam: empty layer");
    TNN_REQUIRE(grads && p && m && v && pows_f64, "tnn_dense_bwd_first_allreduce_adam: NULL argument");
    TNN_REQUIRE(n_params > 0 && n_reduce >= n_params && w_off >= 0 && b_off >= 0 && w_off + n_in * n_out <= n_params &&
                    b_off + n_out <= n_params,
                "tnn_dense_bwd_first_allreduce_adam: the layer's blocks do not lie inside the arena");
    TNN_REQUIRE(!scalar_dst || (scalar_index >= 0 && scalar_index < n_reduce), "tnn_dense_bwd_first_allreduce_adam: scalar_index");
    const size_t esz = dtype == TNN_F64 ? 8 : 4;
    if (dtype == TNN_F32) {
        GemmArgs gw = {};
        gw.A = (const float*)x; gw.B = (const float*)dz; gw.C = nullptr;
        gw.M = n_in; gw.N = n_out; gw.K = rows; gw.lda = n_in; gw.ldb = n_out; gw.ldc = n_out;
        gw.alpha = 1.f; gw.beta = 0.f; gw.epi = EPI_AXPBY;
        const bool aligned = ((reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(p) |
                               reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
        tnn::p2p::LaunchCtx ctx;
        // one launch: the latency kernel's 4-wave form (<= 256 rows per rank), every float4 of the layer's blocks whole and
        // inside one slice, the transport up and large enough
        if (aligned && use_small_path(gw) && !use_mid_path(gw, 1, 0) && rows <= 256 && n_out % 4 == 0 && w_off % 4 == 0 &&
            b_off % 4 == 0 && (!scalar_dst || scalar_index >= n_params) &&
            tnn::p2p_can_allreduce(n_reduce, dtype, TNN_RSUM) && tnn::p2p_launch_ctx(&ctx)) {
            if (int rc = tnn::p2p_refuse_if_failed("tnn_dense_bwd_first_allreduce_adam")) return rc;
            gw.tiles_m = (int)((gw.M + 15) / 16); gw.tiles_n = (int)((gw.N + 15) / 16); gw.splits = 1;
            pick_xcd_cut(gw);
            const int W = ctx.peers.world;
            ArTileArgs f;
            f.buf = (float*)grads; f.n = n_reduce;
            f.slice = ((n_reduce + W - 1) / W + 3) / 4 * 4;
            f.w_off = w_off; f.b_off = b_off;
            f.n_dw = gw.tiles_m * gw.tiles_n;
            tnn::p2p::AdamTail t;
            t.p = (float*)p; t.m = (float*)m; t.v = (float*)v; t.n_params = n_params;
            t.lr = (float)lr; t.b1 = (float)b1; t.b2 = (float)b2; t.eps = (float)eps;
            t.pows = (const double*)pows_f64;
            t.scalar_index = scalar_dst ? scalar_index : -1;
            t.scalar_dst = (float*)scalar_dst;
            // the transport's polling workgroups (128 by default) behind the tiles, 256 threads each: with up to eight ranks'
            // launches on ONE GPU (the tests) they still leave half the wave slots to the producers
            if (dw0_wide_ok(gw)) {                           // 16 x 32 tiles: 392 equal tile workgroups for the MNIST net
                gw.tiles_n = (int)(gw.N / 32);
                pick_xcd_cut(gw);
                f.n_dw = gw.tiles_m * gw.tiles_n;
                hipLaunchKernelGGL((dense_bwd0_allreduce_adam_kernel<4, true>), f.n_dw + ctx.ar_grid, 256, 0, tnn::stream(), gw, f, ctx, t);
            } else {
                hipLaunchKernelGGL((dense_bwd0_allreduce_adam_kernel<4>), f.n_dw + ctx.ar_grid, 256, 0, tnn::stream(), gw, f, ctx, t);
            }
            TNN_LAUNCH_OK();
            return 0;
        }
        // more rows (512 / 1024 per rank): the same single launch on the 32 x 32 tile form of the product
        if (aligned && use_small_path(gw) && use_mid_path(gw, 1, 0) && n_out % 4 == 0 && w_off % 4 == 0 && b_off % 4 == 0 &&
            (!scalar_dst || scalar_index >= n_params) && tnn::p2p_can_allreduce(n_reduce, dtype, TNN_RSUM) && tnn::p2p_launch_ctx(&ctx)) {
            if (int rc = tnn::p2p_refuse_if_failed("tnn_dense_bwd_first_allreduce_adam")) return rc;
            mid_geometry(gw);
            const int W = ctx.peers.world;
            ArTileArgs f;
            f.buf = (float*)grads; f.n = n_reduce;
            f.slice = ((n_reduce + W - 1) / W + 3) / 4 * 4;
            f.w_off = w_off; f.b_off = b_off;
            f.n_dw = gw.tiles_m * gw.tiles_n;
            tnn::p2p::AdamTail t;
            t.p = (float*)p; t.m = (float*)m; t.v = (float*)v; t.n_params = n_params;
            t.lr = (float)lr; t.b1 = (float)b1; t.b2 = (float)b2; t.eps = (float)eps;
            t.pows = (const double*)pows_f64;
            t.scalar_index = scalar_dst ? scalar_index : -1;
            t.scalar_dst = (float*)scalar_dst;
            hipLaunchKernelGGL(dense_bwd0_mid_allreduce_adam_kernel, f.n_dw + ctx.ar_grid, 512, 0, tnn::stream(), gw, f, ctx, t);
            TNN_LAUNCH_OK();
            return 0;
        }
    }
    // anything else: the two launches this replaces
    if (int rc = tnn_dense_bwd(rows, n_in, n_out, x, dz, nullptr, (char*)grads + (size_t)w_off * esz,
                               (char*)grads + (size_t)b_off * esz, nullptr, nullptr, dtype))
        return rc;
    return tnn_allreduce_adam(grads, n_reduce, p, m, v, n_params, lr, b1, b2, eps, const_cast<void*>(pows_f64), 0, dtype,
                              scalar_index, scalar_dst);
}

int tnn_gemm_bias_act(int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A,
                      int64_t lda, const void* B, int64_t ldb, const void* bias, int act,
                      int relu_sign, void* C, int64_t ldc, int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_gemm_bias_act", transA, transB, M, N, K, lda, ldb, ldc)) return rc;
    TNN_REQUIRE(act == TNN_ACT_NONE || act == TNN_ACT_RELU, "tnn_gemm_bias_act: activation %d", act);
    if (M == 0 || N == 0) return 0;
    if (dtype == TNN_F64)
        return gemm_f64(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, 1.0, 0.0, EPI_BIAS_ACT, bias,
                        act, relu_sign, nullptr, 0);
    TNN_REQUIRE(dtype == TNN_F32, "tnn_gemm_bias_act: dtype %d is not a float type", dtype);
    GemmArgs g = {};
    g.A = (const float*)A; g.B = (const float*)B; g.C = (float*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = 1.f; g.beta = 0.f; g.epi = EPI_BIAS_ACT;
    g.bias = (const float*)bias; g.act = act; g.relu_sign = relu_sign;
    return gemm_f32(g, transA, transB);
}

int tnn_dense_fwd_head_partials(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                                           int64_t ldb, const void* bias, int act, int relu_sign, void* C, int64_t ldc,
                                           const void* head_w, int64_t head_c, void* head_z, int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_dense_fwd_head_partials", 0, 0, M, N, K, lda, ldb, ldc)) return rc;
    TNN_REQUIRE(dtype == TNN_F32, "tnn_dense_fwd_head_partials: f32 only (dtype %d)", dtype);
    TNN_REQUIRE(head_w && head_z && head_c >= 1 && head_c <= 16, "tnn_dense_fwd_head_partials: head_w, head_z and 1 <= head_c <= 16");
    TNN_REQUIRE(act == TNN_ACT_NONE || act == TNN_ACT_RELU, "tnn_dense_fwd_head_partials: activation %d", act);
    if (M == 0 || N == 0) return 0;
    GemmArgs g = {};
    g.A = (const float*)A; g.B = (const float*)B; g.C = (float*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = 1.f; g.beta = 0.f; g.epi = EPI_BIAS_ACT;
    g.bias = (const float*)bias; g.act = act; g.relu_sign = relu_sign;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    g.vecA = al(A) && lda % 4 == 0 && K % 4 == 0;
    g.vecB = al(B) && ldb % 4 == 0 && N % 4 == 0;
    if (use_small_path(g) && small_fast_ok(g, 0, 0)) {
        g.head_w = (const float*)head_w; g.head_z = (float*)head_z; g.head_c = (int)head_c;
        return gemm_small(g, 0, 0, nullptr);                   // the partials ride in the tile epilogue: ONE launch
    }
    if (int rc = gemm_f32(g, 0, 0)) return rc;
    const int64_t total = ((N + 15) / 16) * M * head_c;
    hipLaunchKernelGGL(head_partials_kernel, tnn::stream_grid(total, 256), 256, 0, tnn::stream(), (const float*)C, M, N, ldc,
                       (const float*)head_w, (int)head_c, (float*)head_z);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_dense_fwd_head_partials_stats(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                      const void* bias, int act, int relu_sign, void* C, int64_t ldc, const void* head_w,
                                      int64_t head_c, void* head_z, const void* head_b, const void* y, void* ticket_u32,
                                      void* out_pair_f32, int exchange, int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_dense_fwd_head_partials_stats", 0, 0, M, N, K, lda, ldb, ldc)) return rc;
    TNN_REQUIRE(dtype == TNN_F32 && M >= 1 && M <= 1024 && N >= 16 && N <= 256 && N % 16 == 0 && head_c >= 1 && head_c <= 16,
                "tnn_dense_fwd_head_partials_stats: f32, rows <= 1024, hidden width a multiple of 16 up to 256, <= 16 classes");
    TNN_REQUIRE(head_w && head_z && head_b && y && ticket_u32 && out_pair_f32,
                "tnn_dense_fwd_head_partials_stats: head_w, head_z, head_b, y, ticket and out_pair are required");
    TNN_REQUIRE(act == TNN_ACT_NONE || act == TNN_ACT_RELU, "tnn_dense_fwd_head_partials_stats: activation %d", act);
    GemmArgs g = {};
    g.A = (const float*)A; g.B = (const float*)B; g.C = (float*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = 1.f; g.beta = 0.f; g.epi = EPI_BIAS_ACT;
    g.bias = (const float*)bias; g.act = act; g.relu_sign = relu_sign;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    g.vecA = al(A) && lda % 4 == 0 && K % 4 == 0;
    g.vecB = al(B) && ldb % 4 == 0 && N % 4 == 0;
    TNN_REQUIRE(small_fast_ok(g, 0, 0), "tnn_dense_fwd_head_partials_stats: operands must be 16-B aligned with K %% 4 == 0");
    g.head_w = (const float*)head_w; g.head_z = (float*)head_z; g.head_c = (int)head_c;
    g.tiles_m = (int)((M + 15) / 16);
    g.tiles_n = (int)(N / 16);
    g.splits = 1;
    pick_xcd_cut(g);
    tnn::p2p::LaunchCtx ctx = {};
    if (exchange == 2) {
        // DEFERRED exchange (round 6): this launch has no statistics tail at all — the plain forward with its partial logits —
        // and only advances the launch sequence that tags the pairs the head launch behind it exchanges itself
        // (tnn_mlp_head_bwd_tick_xchg).  M <= 128: the head launch's workgroups reduce the shard's pair from the partial logits.
        TNN_REQUIRE(M <= 128, "tnn_dense_fwd_head_partials_stats: exchange = 2 is the <= 128-row form (rows %lld)", (long long)M);
        if (int rc = tnn::p2p_refuse_if_failed("tnn_dense_fwd_head_partials_stats")) return rc;
        TNN_REQUIRE(tnn::p2p_launch_ctx(&ctx), "tnn_dense_fwd_head_partials_stats: the peer-to-peer transport is not enabled");
        g.bump = ctx.xchg_seq;
        return gemm_small(g, 0, 0, nullptr);
    }
    HeadTail ta;
    ta.ticket = (unsigned int*)ticket_u32;
    ta.zpart = (const float*)head_z; ta.bias = (const float*)head_b; ta.y = (const float*)y;
    ta.out_pair = (float*)out_pair_f32; ta.m = (int)M; ta.exchange = exchange ? 1 : 0;
    if (exchange) {
        if (int rc = tnn::p2p_refuse_if_failed("tnn_dense_fwd_head_partials_stats")) return rc;
        TNN_REQUIRE(tnn::p2p_launch_ctx(&ctx), "tnn_dense_fwd_head_partials_stats: the peer-to-peer transport is not enabled");
    }
    if (N == 128 && head_c == 10)
        hipLaunchKernelGGL(dense_fwd_head_kernel<8>, dim3((unsigned)(g.tiles_m * g.tiles_n)), 512, 0, tnn::stream(), g, ta, ctx);
    else
        hipLaunchKernelGGL(dense_fwd_head_generic_kernel<8>, dim3((unsigned)(g.tiles_m * g.tiles_n)), 512, 0, tnn::stream(), g, ta, ctx);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_dense_fwd_rows_head_stats(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                  const void* bias, int act, int relu_sign, void* C, int64_t ldc, const void* head_w,
                                  int64_t head_c, void* head_z_full, const void* head_b, void* pairs_f32, int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_dense_fwd_rows_head_stats", 0, 0, M, N, K, lda, ldb, ldc)) return rc;
    TNN_REQUIRE(dtype == TNN_F32 && M >= 1 && M <= 1024 && N == 128 && head_c == 10 && K >= 1,
                "tnn_dense_fwd_rows_head_stats: f32, 1 <= rows <= 1024, 128 hidden units, 10 classes");
    TNN_REQUIRE(A && B && C && head_w && head_z_full && head_b && pairs_f32,
                "tnn_dense_fwd_rows_head_stats: A, B, C, head_w, head_z_full, head_b and pairs are required");
    TNN_REQUIRE(act == TNN_ACT_RELU, "tnn_dense_fwd_rows_head_stats: the hidden layer in front of the head is a ReLU layer (activation %d)", act);
    TNN_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && lda % 4 == 0 && K % 4 == 0 &&
                    (M - 1) * lda + K < (int64_t)1 << 30 && (K - 1) * ldb + 128 < (int64_t)1 << 30,
                "tnn_dense_fwd_rows_head_stats: A must be 16-B aligned with lda and K multiples of 4, operands below 4 GiB");
    RowPanelArgs g;
    g.A = (const float*)A; g.B = (const float*)B; g.bias = (const float*)bias; g.C = (float*)C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = (int)M; g.K = (int)K; g.relu_sign = relu_sign;
    g.head_w = (const float*)head_w; g.head_b = (const float*)head_b;
    g.zfull = (float*)head_z_full; g.pairs = (float*)pairs_f32; g.bump = nullptr;
    hipLaunchKernelGGL(dense_fwd_rowpanel_head_kernel<false>, dim3((unsigned)((M + 15) / 16)), 512, 0, tnn::stream(), g, HeadTail{},
                       tnn::p2p::LaunchCtx{});
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_dense_fwd_rows_head_stats_merged(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                         const void* bias, int act, int relu_sign, void* C, int64_t ldc, const void* head_w,
                                         int64_t head_c, void* head_z_full, const void* head_b, void* pairs_f32, void* ticket_u32,
                                         void* out_pair_f32, int exchange, int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_dense_fwd_rows_head_stats_merged", 0, 0, M, N, K, lda, ldb, ldc)) return rc;
    TNN_REQUIRE(dtype == TNN_F32 && M >= 1 && M <= 1024 && N == 128 && head_c == 10 && K >= 1,
                "tnn_dense_fwd_rows_head_stats_merged: f32, 1 <= rows <= 1024, 128 hidden units, 10 classes");
    TNN_REQUIRE(A && B && C && head_w && head_z_full && head_b && pairs_f32 && ticket_u32 && out_pair_f32,
                "tnn_dense_fwd_rows_head_stats_merged: A, B, C, head_w, head_z_full, head_b, pairs, ticket and out_pair are required");
    TNN_REQUIRE(act == TNN_ACT_RELU, "tnn_dense_fwd_rows_head_stats_merged: the hidden layer in front of the head is a ReLU layer (activation %d)", act);
    TNN_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && lda % 4 == 0 && K % 4 == 0 &&
                    (M - 1) * lda + K < (int64_t)1 << 30 && (K - 1) * ldb + 128 < (int64_t)1 << 30,
                "tnn_dense_fwd_rows_head_stats_merged: A must be 16-B aligned with lda and K multiples of 4, operands below 4 GiB");
    RowPanelArgs g;
    g.A = (const float*)A; g.B = (const float*)B; g.bias = (const float*)bias; g.C = (float*)C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = (int)M; g.K = (int)K; g.relu_sign = relu_sign;
    g.head_w = (const float*)head_w; g.head_b = (const float*)head_b;
    g.zfull = (float*)head_z_full; g.pairs = (float*)pairs_f32; g.bump = nullptr;
    HeadTail ta = {};
    ta.ticket = (unsigned int*)ticket_u32;
    ta.out_pair = (float*)out_pair_f32; ta.m = (int)M; ta.exchange = exchange ? 1 : 0;
    tnn::p2p::LaunchCtx ctx = {};
    if (exchange == 2) {
        // DEFERRED exchange (round 6): the panels' pairs only — no ticket, no merge, no exchange in this launch; it advances the
        // launch sequence of the exchange the head launch behind it does itself (tnn_mlp_head_bwd_tick_xchg, n_pairs < 0)
        if (int rc = tnn::p2p_refuse_if_failed("tnn_dense_fwd_rows_head_stats_merged")) return rc;
        TNN_REQUIRE(tnn::p2p_launch_ctx(&ctx), "tnn_dense_fwd_rows_head_stats_merged: the peer-to-peer transport is not enabled");
        g.bump = ctx.xchg_seq;
        hipLaunchKernelGGL(dense_fwd_rowpanel_head_kernel<false>, dim3((unsigned)((M + 15) / 16)), 512, 0, tnn::stream(), g, HeadTail{},
                           tnn::p2p::LaunchCtx{});
        TNN_LAUNCH_OK();
        return 0;
    }
    if (exchange) {
        if (int rc = tnn::p2p_refuse_if_failed("tnn_dense_fwd_rows_head_stats_merged")) return rc;
        TNN_REQUIRE(tnn::p2p_launch_ctx(&ctx), "tnn_dense_fwd_rows_head_stats_merged: the peer-to-peer transport is not enabled");
    }
    hipLaunchKernelGGL(dense_fwd_rowpanel_head_kernel<true>, dim3((unsigned)((M + 15) / 16)), 512, 0, tnn::stream(), g, ta, ctx);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_gemm_mask(int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                  const void* B, int64_t ldb, const void* Y, int64_t ldy, void* C, int64_t ldc,
                  int dtype) {
    TNN_NEED_INIT();
    if (int rc = check_shapes("tnn_gemm_mask", transA, transB, M, N, K, lda, ldb, ldc)) return rc;
    TNN_REQUIRE(Y != nullptr && ldy >= N, "tnn_gemm_mask: bad mask operand");
    if (M == 0 || N == 0) return 0;
    if (dtype == TNN_F64)
        return gemm_f64(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, 1.0, 0.0, EPI_MASK, nullptr, 0,
                        0, Y, ldy);
    TNN_REQUIRE(dtype == TNN_F32, "tnn_gemm_mask: dtype %d is not a float type", dtype);
    GemmArgs g = {};
    g.A = (const float*)A; g.B = (const float*)B; g.C = (float*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = 1.f; g.beta = 0.f; g.epi = EPI_MASK;
    g.Y = (const float*)Y; g.ldy = ldy;
    return gemm_f32(g, transA, transB);
}

}  // extern "C"
