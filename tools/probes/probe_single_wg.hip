// Probe: what do the phases of a single-workgroup kernel cost on MI355X when launched back-to-back from a
// hipGraph?  Prints shader-clock cycles (s_memtime) and 100 MHz wall ticks per phase, plus event time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void probe(const float* __restrict__ in, float* __restrict__ out, int n,
                                               unsigned long long* stamps, int nbar, int fma_iters) {
    __shared__ float lds[8192];
    unsigned long long t[8], w[8];
    int k = 0;
    t[k] = __builtin_readcyclecounter(); w[k++] = wall_clock64();
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { float v = in[i]; lds[i & 8191] = v; acc += v; }   // global load
    t[k] = __builtin_readcyclecounter(); w[k++] = wall_clock64();
    for (int b = 0; b < nbar; ++b) __syncthreads();                                                       // barriers
    t[k] = __builtin_readcyclecounter(); w[k++] = wall_clock64();
    float x = acc;
    for (int i = 0; i < fma_iters; ++i) x = fmaf(x, 1.0001f, 0.5f);                                        // dependent VALU chain
    t[k] = __builtin_readcyclecounter(); w[k++] = wall_clock64();
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = x + lds[(i * 7) & 8191];                   // store
    t[k] = __builtin_readcyclecounter(); w[k++] = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamps)
        for (int i = 0; i < k; ++i) { stamps[2 * i] = t[i]; stamps[2 * i + 1] = w[i]; }
}

int main() {
    const int n = 8192;
    float *in, *out; unsigned long long* st;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&st, 128));
    CK(hipMemset(in, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Cfg { int threads, nbar, fma; const char* name; } cfgs[] = {
        {1024, 0, 0, "1024 thr: load+store only"}, {1024, 10, 0, "1024 thr: +10 barriers"},
        {1024, 10, 1000, "1024 thr: +10 barriers +1000 dep FMA"}, {256, 10, 1000, "256 thr: same"},
        {64, 0, 0, "64 thr: load+store only"}};
    for (auto& c : cfgs) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe, 1, c.threads, 0, s, in, out, n, st, c.nbar, c.fma);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[16]; CK(hipMemcpy(h, st, 128, hipMemcpyDeviceToHost));
        printf("%-40s %6.2f us/launch | cycles: load %llu bar %llu fma %llu store %llu | wall(10ns): %llu %llu %llu %llu\n",
               c.name, ms * 1000 / 200, h[2] - h[0], h[4] - h[2], h[6] - h[4], h[8] - h[6],
               h[3] - h[1], h[5] - h[3], h[7] - h[5], h[9] - h[7]);
    }
    return 0;
}
