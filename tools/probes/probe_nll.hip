// Phase timing of the single-block whole-batch softmax-NLL kernel (copy of nll_fused_kernel<float> with
// s_memtime stamps) to see which phase costs what at m=128, c=10.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool IS_MAX, typename R>
__device__ __forceinline__ double block_reduce_fast(R v, R* slots) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { R other = __shfl_xor(v, o, 64); v = IS_MAX ? (other > v ? other : v) : v + other; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) slots[w] = v;
    __syncthreads();
    double r = (double)slots[0];
    for (int i = 1; i < nw; ++i) { const double x = (double)slots[i]; r = IS_MAX ? (x > r ? x : r) : r + x; }
    return r;
}

__global__ __launch_bounds__(1024) void nll(const float* __restrict__ z, const float* __restrict__ y, int m, int c,
                                            float* stats_out, float* loss_out, float* dz, unsigned long long* st) {
    __shared__ float e_lds[4096], y_lds[4096];
    __shared__ double q_lds[1024];
    __shared__ float red_max[16], red_sum[16], red_loss[16];
    __shared__ double scal[2];
    unsigned long long t[10]; int k = 0;
    t[k++] = __builtin_readcyclecounter();
    const int tid = threadIdx.x, n = m * c;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += blockDim.x) { const float zi = z[i]; y_lds[i] = y[i]; e_lds[i] = zi; mx = zi > mx ? zi : mx; }
    t[k++] = __builtin_readcyclecounter();
    const double M = block_reduce_fast<true, float>(mx, red_max);
    t[k++] = __builtin_readcyclecounter();
    const float Mt = (float)M;
    double s = 0.0;
    for (int i = tid; i < n; i += blockDim.x) { const double e = (double)expf(e_lds[i] - Mt); e_lds[i] = (float)e; y_lds[i] = (float)((double)(float)e * (double)y_lds[i]); s += e; }
    t[k++] = __builtin_readcyclecounter();
    const double S = block_reduce_fast<false, float>((float)s, red_sum);
    t[k++] = __builtin_readcyclecounter();
    if (tid == 0) { scal[0] = log(S); scal[1] = 1.0 / S; }
    const double inv_m = 1.0 / (double)m;
    double local = 0.0;
    for (int r = tid; r < m; r += blockDim.x) {
        double q = 0.0;
        for (int kk = 0; kk < c; ++kk) q += (double)y_lds[r * c + kk];
        q_lds[r] = inv_m / q;
        local -= (double)logf((float)q);
    }
    t[k++] = __builtin_readcyclecounter();
    const double sum_log_q = block_reduce_fast<false, float>((float)local, red_loss);
    t[k++] = __builtin_readcyclecounter();
    const double loss = scal[0] + sum_log_q * inv_m;
    const float inv_sf = (float)scal[1];
    for (int i = tid; i < n; i += blockDim.x) dz[i] = e_lds[i] * inv_sf - y_lds[i] * (float)q_lds[i / c];
    t[k++] = __builtin_readcyclecounter();
    if (tid == 0) { loss_out[0] = (float)loss; stats_out[0] = (float)M; stats_out[1] = (float)S; for (int i = 0; i < k; ++i) st[i] = t[i]; }
}

int main() {
    const int m = 128, c = 10, n = m * c;
    float *z, *y, *dz, *stats, *loss; unsigned long long* st;
    CK(hipMalloc(&z, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&dz, n * 4)); CK(hipMalloc(&stats, 8)); CK(hipMalloc(&loss, 4)); CK(hipMalloc(&st, 128));
    float hz[n], hy[n]; for (int i = 0; i < n; ++i) { hz[i] = (i % 7) * 0.1f; hy[i] = (i % c) == (i / c) % c; }
    CK(hipMemcpy(z, hz, n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(y, hy, n * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int threads : {1024, 512, 256}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(nll, 1, threads, 0, s, z, y, m, c, stats, loss, dz, st);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[16]; CK(hipMemcpy(h, st, 128, hipMemcpyDeviceToHost));
        printf("%4d thr %6.2f us/launch | cycles: load %llu max %llu exp %llu sum %llu rows %llu loss %llu dz %llu | total %llu\n", threads,
               ms * 1000 / 200, h[1]-h[0], h[2]-h[1], h[3]-h[2], h[4]-h[3], h[5]-h[4], h[6]-h[5], h[7]-h[6], h[7]-h[0]);
    }
    return 0;
}
