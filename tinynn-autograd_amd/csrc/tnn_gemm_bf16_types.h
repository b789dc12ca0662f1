// Types, the argument block and small device helpers shared by the bf16 GEMM kernels (tnn_gemm_bf16.hip, tnn_gemm_bf16_dma.h,
// tnn_gemm_bf16_sk.h).  Included INSIDE the including file's anonymous namespace (the kernels have internal linkage), by the
// library's translation unit and by the stand-alone probe tools/probes/gemm_bf16_sk_probe.hip.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint16_t bf16_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native 16-B vector (HIP's uint4 struct ended up in scratch)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { BEPI_PLAIN = 0, BEPI_BIAS_ACT = 1, BEPI_MASK = 2 };

struct BfArgs {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    int64_t M, N, K, lda, ldb, ldc;
    int c_bf16;            // output element type: 1 = bf16, 0 = f32
    int epi;
    const float* bias;
    int act, relu_sign;
    const bf16_t* Y;
    int64_t ldy;
    int tiles_m, tiles_n;
    // BEPI_ADAM (tnn_gemm_bf16_nt_adam): the product is a weight gradient that Adam consumes in the epilogue
    float *ap, *am, *av;             // fp32 master weights and moments, [M][ldc] like C
    bf16_t *aw16, *awT16;            // bf16 working copy [M][ldc] and its transpose [N][ldt]
    int64_t ldt;
    float lr, b1, b2, eps;
    const double* pows;              // {b1^t, b2^t}, already advanced for this step
    const int* guard;                // data-parallel update guard (tnn_internal.h)
    // split-K kernel (tnn_gemm_bf16_sk.h): K slices per output tile, fp32 partial slabs [tile][slice][32][512] float4 and
    // two flag words per tile (they count launches, never reset)
    int splitk;
    float* sk_ws;
    unsigned* sk_cnt;
    unsigned long long* sk_trace;   // probe builds only: per-workgroup timestamps
    // optional second output of the bf16 epilogues: the TRANSPOSE of C, CT [N][ldct] (element (m, n) at CT[n * ldct + m]) —
    // the K-contiguous operand the dW product of the backward pass wants (core/ops.py:159-160), NULL = not wanted
    bf16_t* CT;
    int64_t ldct;
    int* fault;                     // the process's sticky fault word (tnn::fault_word), NULL in probe builds
};

__device__ __forceinline__ bf16_t f2bf(float f) {       // round to nearest even (finite inputs)
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two floats -> packed bf16 pair (a in the low half), round to nearest even: the __bf16 conversion compiles to ONE
// v_cvt_pk_bf16_f32 on gfx950 where the integer form above costs ~6 VALU instructions per element (bit-identical for finite
// inputs; NaNs come out quiet)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }

__device__ __forceinline__ int xcd_remap16(int b, int nb) {
    const int nx = 8;
    if (nb < 2 * nx) return b;
    int q = nb / nx, r = nb % nx;
    int xcd = b % nx, local = b / nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

