// libtnn_probe.so — tnn_probe_box: what THIS box can do, measured in ~100 ms — so that a roofline fraction can be normalised by the box it was
// measured on (box-to-box spread of the MFMA- and HBM-bound numbers is +-4..8 %):
//   * MFMA-only loops (nothing but v_mfma on register operands, 8 waves per CU, 8 independent accumulators per wave):
//     fp32 (32x32x2_f32) and bf16 (32x32x16_bf16) with uniform-random operands and, for bf16, with zeros — the chip clocks to
//     its power budget, so random data is the number a GEMM on real data can approach (1.6-1.8 of the 2.5 PFLOP/s on this
//     pool) and zeros show the clock-unconstrained pipe;
//   * the sustained shader clock of each loop: s_memtime (shader cycles) over s_memrealtime (100 MHz) inside the kernel;
//   * HBM streaming past the 256 MB memory-side cache, bytes read + written per second: a float4 copy (1 GiB + 1 GiB) and
//     the optimizer's mix (four 256 MB arrays read, three of them rewritten in place).
// No reference counterpart: measurement infrastructure for bench.py's `box` object, a library of its OWN (not part of the
// product's libtnn_hip.so, no symbol of include/tnn_hip.h): own stream, own buffers, plain int return codes.
//   make -C tinynn-autograd_amd/csrc probe      ->  tools/probes/bin/libtnn_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// clocks[2 * block] = shader cycles, clocks[2 * block + 1] = 100 MHz ticks spent in the loop (thread 0 of each workgroup)
// zero: 0 = operands uniform in [-1, 1), 1 = zeros, 2 (fp32) = the training step's distribution: a uniform in [0, 1)
// (activations), b uniform in +-0.027 (Xavier weights of a 4096-wide layer)
template <bool BF16>
__global__ __launch_bounds__(512) void mfma_only_kernel(int iters, int zero, float* sink, unsigned long long* clocks) {
    const uint32_t seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (BF16) {
        u32x4 raw[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // two bf16 values uniform in [-1, 1): sign + exponent of 1.x minus 1 is awkward in bits — take a float, round
                const uint32_t h = hash32(seed + 16u * i + e);
                const float f0 = (float)(h & 0xffffu) * (2.0f / 65536.0f) - 1.0f, f1 = (float)(h >> 16) * (2.0f / 65536.0f) - 1.0f;
                raw[i][e] = zero ? 0u : ((__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u));
            }
        bf16x8 a[2], b[4];
        a[0] = __builtin_bit_cast(bf16x8, raw[0]); a[1] = __builtin_bit_cast(bf16x8, raw[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8, raw[2 + i]);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
        }
    } else {
        float a[2], b[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float u = (float)(hash32(seed + i) >> 8) * (1.0f / 16777216.0f);          // [0, 1)
            a[i] = zero == 1 ? 0.f : zero == 2 ? u : 2.f * u - 1.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u = (float)(hash32(seed + 2 + i) >> 8) * (1.0f / 16777216.0f);
            b[i] = zero == 1 ? 0.f : (zero == 2 ? 0.027f : 1.f) * (2.f * u - 1.f);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(acc[i]));        // the loop's MFMAs have retired
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        clocks[2 * blockIdx.x] = c1 - c0;
        clocks[2 * blockIdx.x + 1] = t1 - t0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
// the optimizer's stream mix: four arrays read, three of them rewritten IN PLACE (16 B + 12 B per element; a write that
// follows the read of the same line finds its DRAM page open — writing to separate arrays measured 5.4 TB/s where this form
// and the fused Adam reach 6.3-6.4), trivial arithmetic
__global__ __launch_bounds__(256) void mix43_kernel(f32x4* __restrict__ a, f32x4* __restrict__ b, f32x4* __restrict__ c,
                                                    const f32x4* __restrict__ d, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        // the fused Adam's cache policies: the gradient and the moments non-temporal, the parameters ordinary
        const f32x4 vd = __builtin_nontemporal_load(d + i), vb = __builtin_nontemporal_load(b + i),
                    vc = __builtin_nontemporal_load(c + i), va = a[i];
        __builtin_nontemporal_store(vb - vd, b + i);
        __builtin_nontemporal_store(vc + vd, c + i);
        a[i] = va + vd;
    }
}

}  // namespace


#define PB_CHECK(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess) {                                                                             \
            fprintf(stderr, "tnn_probe_box: %s -> %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return 1;                                                                                        \
        }                                                                                                    \
    } while (0)

static unsigned stream_grid(int64_t items, int cus) {
    int64_t b = (items + 255) / 256, cap = (int64_t)cus * 8;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

/* out[0] fp32 MFMA-only TFLOP/s (v_mfma_f32_32x32x2_f32, random operands), out[1] its sustained shader clock in GHz,
 * out[2] / out[3] bf16 (v_mfma_f32_32x32x16_bf16) with random operands, out[4] / out[5] with zero operands (the chip clocks to
 * its power budget), out[6] float4 copy bandwidth in GB/s (1 GiB read + 1 GiB written), out[7] GB/s of the optimizer's stream
 * mix (four 256 MB arrays read, three of them rewritten in place), out[8] / out[9] the fp32 loop with the training step's operand
 * distribution (a uniform in [0, 1), b uniform in +-0.027).  n_out >= 10.  Uses the CURRENT HIP device. */
extern "C" __attribute__((visibility("default"))) int tnn_probe_box(double* out, int n_out) {
    if (out == nullptr || n_out < 10) return 2;
    int dev = 0;
    PB_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PB_CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipStream_t s;
    PB_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    void *sink = nullptr, *clocks = nullptr;
    PB_CHECK(hipMalloc(&sink, 4096));
    PB_CHECK(hipMalloc(&clocks, (size_t)cus * 16));
    hipEvent_t e0, e1;
    PB_CHECK(hipEventCreate(&e0));
    PB_CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> hc((size_t)cus * 2);
    auto mfma = [&](bool bf, int zero, int iters, double flop_per_mfma, double* tflops, double* ghz) -> int {
        for (int pass = 0; pass < 2; ++pass) {            // pass 0 warms up (clock ramp), pass 1 is timed
            PB_CHECK(hipEventRecord(e0, s));
            if (bf) hipLaunchKernelGGL(mfma_only_kernel<true>, dim3(cus), 512, 0, s, iters, zero, (float*)sink, (unsigned long long*)clocks);
            else hipLaunchKernelGGL(mfma_only_kernel<false>, dim3(cus), 512, 0, s, iters, zero, (float*)sink, (unsigned long long*)clocks);
            PB_CHECK(hipEventRecord(e1, s));
            PB_CHECK(hipEventSynchronize(e1));
        }
        float ms = 0.f;
        PB_CHECK(hipEventElapsedTime(&ms, e0, e1));
        PB_CHECK(hipMemcpy(hc.data(), clocks, hc.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, ticks = 0;
        for (int b = 0; b < cus; ++b) { cyc += (double)hc[2 * b]; ticks += (double)hc[2 * b + 1]; }
        *tflops = (double)cus * 8 * iters * 32.0 * flop_per_mfma / (ms * 1e-3) / 1e12;
        *ghz = ticks > 0 ? cyc / (ticks * 10.0) : 0.0;    // cycles per 10 ns tick -> GHz
        return 0;
    };
    int rc = 0;
    rc = rc ? rc : mfma(false, 0, 2048, 2.0 * 32 * 32 * 2, &out[0], &out[1]);        // fp32, random operands (~14 ms)
    rc = rc ? rc : mfma(true, 0, 4096, 2.0 * 32 * 32 * 16, &out[2], &out[3]);        // bf16, random operands (~20 ms)
    rc = rc ? rc : mfma(true, 1, 4096, 2.0 * 32 * 32 * 16, &out[4], &out[5]);        // bf16, zeros
    rc = rc ? rc : mfma(false, 2, 2048, 2.0 * 32 * 32 * 2, &out[8], &out[9]);        // fp32, the step's operand distribution
    // float4 copy, 1 GiB each way
    const int64_t bytes = (int64_t)1 << 30;
    void *a = nullptr, *b = nullptr;
    if (!rc && (hipMalloc(&a, (size_t)bytes) != hipSuccess || hipMalloc(&b, (size_t)bytes) != hipSuccess)) rc = 1;
    if (!rc) {
        PB_CHECK(hipMemsetAsync(a, 0x3c, (size_t)bytes, s));
        for (int pass = 0; pass < 2; ++pass) {
            PB_CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(copy16_kernel, dim3(stream_grid(bytes / 16, cus)), 256, 0, s, (const f32x4*)a, (f32x4*)b, bytes / 16);
            PB_CHECK(hipEventRecord(e1, s));
            PB_CHECK(hipEventSynchronize(e1));
        }
        float ms = 0.f;
        PB_CHECK(hipEventElapsedTime(&ms, e0, e1));
        out[6] = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;      // GB/s, read + written
    }
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    // the optimizer's mix over 4 x 256 MB
    const int64_t mb = (int64_t)256 << 20;
    void* arr[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4 && !rc; ++i)
        if (hipMalloc(&arr[i], (size_t)mb) != hipSuccess) rc = 1;
    if (!rc) {
        for (int i = 0; i < 4; ++i) PB_CHECK(hipMemsetAsync(arr[i], 0x3c, (size_t)mb, s));
        for (int pass = 0; pass < 2; ++pass) {
            PB_CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(mix43_kernel, dim3(stream_grid(mb / 16, cus)), 256, 0, s, (f32x4*)arr[0], (f32x4*)arr[1],
                               (f32x4*)arr[2], (const f32x4*)arr[3], mb / 16);
            PB_CHECK(hipEventRecord(e1, s));
            PB_CHECK(hipEventSynchronize(e1));
        }
        float ms = 0.f;
        PB_CHECK(hipEventElapsedTime(&ms, e0, e1));
        out[7] = 7.0 * (double)mb / (ms * 1e-3) / 1e9;         // GB/s, 4 streams read + 3 written
    }
    for (int i = 0; i < 4; ++i)
        if (arr[i]) (void)hipFree(arr[i]);
    (void)hipFree(sink);
    (void)hipFree(clocks);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(s);
    PB_CHECK(hipGetLastError());
    return rc;
}
