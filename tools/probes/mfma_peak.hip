// Attainable fp32 MFMA rate on this chip: nothing but v_mfma_f32_32x32x2_f32 on 1, 2 or 3 waves per SIMD,
// with 2 or 4 independent accumulators per wave.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NACC>
void run(int wgs_per_cu) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    float* out;
    hipMalloc(&out, 4096);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<cus * wgs_per_cu, 256>>>(out, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<cus * wgs_per_cu, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)cus * wgs_per_cu * 4 * iters * 8.0 * NACC * 4096.0;
    printf("acc %d, %d waves/SIMD: %.2f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, ms, flops / ms / 1e9);
}

int main() {
    run<2>(1); run<2>(2); run<4>(1); run<4>(2); run<2>(3);
    return 0;
}
