// Stand-alone probe of the skinny bf16 GEMM (512 x 8192 x 8192 -> bf16, config E's forward / dX shape): the split-K
// 256-row-tile kernel in several geometries + timing-only ablation builds (gemm_bf16_sk_probe_kernel.h; the library ships only
// the chosen form, csrc/tnn_gemm_bf16_sk.h), against the library's
// 128 x 128 LDS-DMA kernel and a naive reference product.  Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I tinynn-autograd_amd/csrc -I tools/probes tools/probes/gemm_bf16_sk_probe.hip \
//         -o tools/probes/bin/gemm_bf16_sk_probe && tools/probes/bin/gemm_bf16_sk_probe
// Weights rotate over three matrices (402 MB > the 256 MB memory-side cache) so every call streams them from HBM, like
// the training step does.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <type_traits>
#include <vector>

#include "tnn_hip.h"

namespace {
#include "tnn_gemm_bf16_types.h"
#include "tnn_gemm_bf16_dma.h"
#include "gemm_bf16_sk_probe_kernel.h"   // the probe's OWN copy: every geometry / hand-off protocol / ablation switch tried

__global__ void ref_kernel(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(bf2f(A[(int64_t)m * K + k]), bf2f(B[(int64_t)n * K + k]), s);
    C[(int64_t)m * N + n] = s;
}

__global__ void fill_kernel(bf16_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;      // uniform [-1, 1)
        p[i] = f2bf(f);
    }
}

// attainable bf16 MFMA rate at this kernel's geometry (8 waves per CU, 8 independent 32x32x16 accumulators per wave), operands
// taken from memory once: random data vs zeros (the chip clocks to its power budget: guide, DVFS)
__global__ __launch_bounds__(512) void mfma_peak_kernel(const bf16_t* src, float* out, int iters) {
    const int tid = threadIdx.x;
    bf16x8 a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + ((blockIdx.x * 512 + tid) * 6 + i) * 8));
    for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + ((blockIdx.x * 512 + tid) * 6 + 2 + i) * 8));
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[tid] = s;
}

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e__ = (x);                                                              \
        if (e__ != hipSuccess) {                                                           \
            fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e__), __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct Ctx {
    int M = 512, N = 8192, K = 8192;
    bf16_t* A = nullptr;
    std::vector<bf16_t*> B;
    std::vector<bf16_t*> Bt;             // the same matrices stored [K][N] (n-contiguous) for the BNC variants
    bf16_t* C = nullptr;
    float* ref = nullptr;
    float* ws = nullptr;
    unsigned* cnt = nullptr;
    unsigned long long* trace = nullptr;
    std::vector<uint16_t> h_ref16;       // reference rounded to bf16 (for B[0])
    std::vector<float> h_ref, h_ref1;    // A x B[0]^T, A x B[1]^T
};

BfArgs base_args(const Ctx& c, int b) {
    BfArgs g = {};
    g.A = c.A; g.B = c.B[b]; g.C = c.C;
    g.M = c.M; g.N = c.N; g.K = c.K; g.lda = c.K; g.ldb = c.K; g.ldc = c.N;
    g.c_bf16 = 1;
    g.epi = BEPI_PLAIN;
    g.sk_ws = c.ws;
    g.sk_cnt = c.cnt;
    g.sk_trace = c.trace;
    return g;
}

struct Variant { std::string name; std::function<void(int)> launch; };
std::vector<Variant> g_variants;
bool g_solo = false;       // also time every variant on its own, back to back (chip state drifts between variants)

// Every variant in turn, 12 launches each, for `rounds` rounds: the chip clocks to its power / thermal budget and a kernel timed
// alone right after start-up runs 20-30 % faster than the same kernel in a sustained mix, so variants are compared INSIDE one
// sustained state: median over the rounds.
void run_interleaved(Ctx& c, int rounds) {
    const int reps = 12, nv = (int)g_variants.size();
    std::vector<std::vector<float>> t(nv);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds + 1; ++r)
        for (int v = 0; v < nv; ++v) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) g_variants[v].launch(i % (int)c.B.size());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) t[v].push_back(ms / reps * 1e3f);      // round 0 = warm-up
        }
    const double fl = 2.0 * c.M * c.N * c.K;
    printf("---- interleaved, %d rounds x %d launches per variant: median / min / max us, TFLOP/s at the median\n", rounds, reps);
    for (int v = 0; v < nv; ++v) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2];
        printf("%-48s %7.1f %7.1f %7.1f   %7.1f\n", g_variants[v].name.c_str(), med, t[v].front(), t[v].back(), fl / med / 1e6);
    }
}

template <typename F>
void run_variant(const char* name, Ctx& c, F launch, bool check) {
    hipStream_t s = 0;
    if (check) {
        CK(hipMemsetAsync(c.C, 0xff, (size_t)c.M * c.N * 2, s));
        launch(0);
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> out((size_t)c.M * c.N), out2((size_t)c.M * c.N);
        CK(hipMemcpy(out.data(), c.C, out.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        int64_t bad = 0;
        for (size_t i = 0; i < out.size(); ++i) {
            const float got = __builtin_bit_cast(float, (uint32_t)out[i] << 16), want = c.h_ref[i];
            const double err = fabs((double)got - want), tol = 0.01 * fabs(want) + 0.05;   // bf16 rounding of |x| <~ 60
            if (!(err <= tol)) ++bad;
            if (err > worst) worst = err;
        }
        // determinism + no stale hand-off data: launches alternate between two weight matrices (a slab or an output that
        // survived from the previous launch would belong to the OTHER product), every result bit-identical to the first of
        // its kind and inside the tolerance of its own reference
        int64_t nondet = 0, bad1 = 0;
        std::vector<uint16_t> first1;
        for (int r = 0; r < 8 && c.B.size() > 1; ++r) {
            const int which = (r + 1) & 1;
            launch(which);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(out2.data(), c.C, out2.size() * 2, hipMemcpyDeviceToHost));
            if (which == 0) {
                if (memcmp(out.data(), out2.data(), out.size() * 2) != 0) ++nondet;
            } else {
                if (first1.empty()) {
                    first1 = out2;
                    for (size_t i = 0; i < out2.size(); ++i) {
                        const float got = __builtin_bit_cast(float, (uint32_t)out2[i] << 16), want = c.h_ref1[i];
                        if (!(fabs((double)got - want) <= 0.01 * fabs(want) + 0.05)) ++bad1;
                    }
                } else if (memcmp(first1.data(), out2.data(), out2.size() * 2) != 0) ++nondet;
            }
        }
        printf("%-44s check: max abs err %.4f, %lld + %lld outside tolerance, %lld of 8 alternating reruns differ\n", name, worst,
               (long long)bad, (long long)bad1, (long long)nondet);
    }
    g_variants.push_back({name, launch});
    if (!g_solo) return;
    for (int i = 0; i < 3; ++i) launch(i % (int)c.B.size());
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const int rounds = 5, reps = 12;
    float per_round[rounds];
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch(i % (int)c.B.size());
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        best = fminf(best, ms);
        sum += ms;
        per_round[r] = ms;
    }
    const double fl = 2.0 * c.M * c.N * c.K;
    printf("%-44s %8.1f us avg  %8.1f us best  %7.1f TFLOP/s (best)   rounds:", name, sum / rounds * 1e3, best * 1e3, fl / best / 1e9);
    for (int r = 0; r < rounds; ++r) printf(" %.1f", per_round[r] * 1e3);
    printf("\n");
    fflush(stdout);
}

__global__ void transpose_probe_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols) {
    // dst[c][r] = src[r][c] (set-up only)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)rows * cols) { const int r = (int)(i / cols), cc = (int)(i % cols); dst[(int64_t)cc * rows + r] = src[i]; }
}

template <int BN, int WM, int WN, int NSA, int NSB, int S, int ABL, int SYM = 0, int BNC = 0, int BMT = 256>
void sk_variant(const char* name, Ctx& c, bool check) {
    auto launch = [&](int b) {
        BfArgs g = base_args(c, b);
        if (BNC == 1 || BNC == 2 || BNC == 5) { g.B = c.Bt[b]; g.ldb = c.N; }
        g.tiles_m = (c.M + BMT - 1) / BMT;
        g.tiles_n = (c.N + BN - 1) / BN;
        g.splitk = S;
        if (SYM) g.sk_cnt = c.cnt + (BMT == 256 ? 2048 : 3072);      // the symmetric hand-off's flags count launches (a set per geometry); the ticket protocol's words return to zero
        hipLaunchKernelGGL((sk::gemm_bf16_sk_kernel<BN, WM, WN, NSA, NSB, S, ABL, SYM, BNC, BMT>), dim3(g.tiles_m * g.tiles_n * S), 512, 0, 0, g);
    };
    run_variant(name, c, launch, check && ABL == 0);
    if constexpr ((ABL & 32) != 0) {
        for (int i = 0; i < 4; ++i) launch(i % (int)c.B.size());
        CK(hipDeviceSynchronize());
        // the last launch's timestamps: medians over the workgroups of {prologue, K loop, tail} in shader cycles and in us
        const int nb = ((c.M + BMT - 1) / BMT) * ((c.N + BN - 1) / BN) * S;
        std::vector<unsigned long long> t((size_t)nb * 10);
        CK(hipMemcpy(t.data(), c.trace, t.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> pro_u, loop_c, loop_u, xch_u, epi_u, start_u, end_u;
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb; ++b) t0 = t[10 * b + 1] < t0 ? t[10 * b + 1] : t0;
        for (int b = 0; b < nb; ++b) {
            const unsigned long long* q = &t[10 * b];
            pro_u.push_back((q[3] - q[1]) * 0.01);
            loop_c.push_back((double)(q[4] - q[2]));
            loop_u.push_back((q[5] - q[3]) * 0.01);
            if (q[9]) {                                   // the workgroup ran the final epilogue (reducers; everyone without exchange)
                xch_u.push_back((q[9] - q[5]) * 0.01);
                epi_u.push_back((q[7] - q[9]) * 0.01);
            }
            start_u.push_back((q[1] - t0) * 0.01);
            end_u.push_back((q[7] - t0) * 0.01);
        }
        auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v.back(); };
        printf("    trace (medians, %d workgroups): prologue %.2f us | K loop %.0f cyc = %.2f us (%.2f GHz) | exchange of the %d finishing "
               "workgroups %.2f us (max %.2f) | their epilogue %.2f us (max %.2f) | end of kernel: median %.2f, last %.2f us\n",
               nb, med(pro_u), med(loop_c), med(loop_u), med(loop_c) / med(loop_u) * 1e-3, (int)xch_u.size(), med(xch_u), mx(xch_u),
               med(epi_u), mx(epi_u), med(end_u), mx(end_u));
    }
}

}  // namespace

int main(int argc, char** argv) {
    Ctx c;
    if (argc > 3) { c.M = atoi(argv[1]); c.N = atoi(argv[2]); c.K = atoi(argv[3]); }
    const int NB = (argc > 4 && !strcmp(argv[4], "norotate")) ? 1 : 3;
    CK(hipMalloc(&c.A, (size_t)c.M * c.K * 2));
    c.B.resize(NB);
    for (int i = 0; i < NB; ++i) CK(hipMalloc(&c.B[i], (size_t)c.N * c.K * 2));
    CK(hipMalloc(&c.C, (size_t)c.M * c.N * 2));
    CK(hipMalloc(&c.ref, (size_t)c.M * c.N * 4));
    const size_t ws_bytes = (size_t)((c.M + 255) / 256) * 256 * (size_t)c.N * 4 * 4 * 2;   // up to S = 4, BN = 128 edge slack
    CK(hipMalloc(&c.ws, ws_bytes));
    CK(hipMalloc(&c.cnt, 4096 * 4));
    CK(hipMemset(c.cnt, 0, 4096 * 4));
    CK(hipMalloc(&c.trace, 4096 * 10 * 8));
    fill_kernel<<<2048, 256>>>(c.A, (int64_t)c.M * c.K, 0x1234u);
    for (int i = 0; i < NB; ++i) fill_kernel<<<2048, 256>>>(c.B[i], (int64_t)c.N * c.K, 0x9e37u + 77u * i);
    c.Bt.resize(NB);
    for (int i = 0; i < NB; ++i) {
        CK(hipMalloc(&c.Bt[i], (size_t)c.N * c.K * 2));
        transpose_probe_kernel<<<(unsigned)(((int64_t)c.N * c.K + 255) / 256), 256>>>(c.B[i], c.Bt[i], c.N, c.K);
    }
    ref_kernel<<<dim3((c.N + 255) / 256, c.M), 256>>>(c.A, c.B[0], c.ref, c.M, c.N, c.K);
    CK(hipDeviceSynchronize());
    c.h_ref.resize((size_t)c.M * c.N);
    CK(hipMemcpy(c.h_ref.data(), c.ref, c.h_ref.size() * 4, hipMemcpyDeviceToHost));
    if (NB > 1) {
        ref_kernel<<<dim3((c.N + 255) / 256, c.M), 256>>>(c.A, c.B[1], c.ref, c.M, c.N, c.K);
        CK(hipDeviceSynchronize());
        c.h_ref1.resize((size_t)c.M * c.N);
        CK(hipMemcpy(c.h_ref1.data(), c.ref, c.h_ref1.size() * 4, hipMemcpyDeviceToHost));
    }
    printf("shape %d x %d x %d, reference[0..3] = %.4f %.4f %.4f %.4f\n", c.M, c.N, c.K, c.h_ref[0], c.h_ref[1], c.h_ref[2], c.h_ref[3]);

    for (int zero = 0; zero < 4; ++zero) {
        const int iters = zero >= 2 ? 32 : 256;      // 256 x 32 MFMAs per wave = 8192; 32 x 32 = the GEMM's 1024 per wave
        const int z2 = zero;
        zero &= 1;
        bf16_t* src = c.B[0];
        if (zero) CK(hipMemset(c.ws, 0, (size_t)32 << 20));
        const bf16_t* p = zero ? (const bf16_t*)c.ws : src;         // 12.6 MB are read
        mfma_peak_kernel<<<256, 512>>>(p, c.ref, iters);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; ++r) mfma_peak_kernel<<<256, 512>>>(p, c.ref, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        const double fl = 256.0 * 8 * iters * 32 * (2.0 * 32 * 32 * 16);
        printf("MFMA-only micro-benchmark, %s operands, %d MFMAs per wave: %.1f us, %.0f TFLOP/s\n", zero ? "zero" : "random", iters * 32,
               ms * 1e3, fl / ms / 1e9);
        zero = z2;
    }

    {   // the library's kernel for this shape today: 128 x 128 tiles, 8 waves, 4-stage ring, one tile per CU
        auto launch = [&](int b) {
            BfArgs g = base_args(c, b);
            g.tiles_m = (c.M + 127) / 128;
            g.tiles_n = (c.N + 127) / 128;
            hipLaunchKernelGGL((gemm_bf16_dma_kernel<8, 4, true>), dim3(g.tiles_m * g.tiles_n), 512, 0, 0, g);
        };
        run_variant("baseline 128x128 dma8 (4 stages)", c, launch, true);
    }
    //            BN  WM WN NSA NSB S ABL          ABL bits: 1 no MFMA, 2 no refill DMA, 4 no exchange, 8 no frag reads, 16 no barrier, 32 trace,
    //                                              64 B addressed tile-major (timing only), 128 one K-tile per loop trip
    sk_variant<128, 4, 2, 3, 4, 2, 0, 1>("sk 256x128 S2 A3 B4 symmetric hand-off", c, true);
    sk_variant<128, 4, 2, 3, 4, 2, 0, 1, 1>("sk 256x128 S2 A3 B4 symmetric, B n-contiguous (tr reads)", c, true);
    sk_variant<128, 4, 2, 3, 4, 2, 0, 1, 2>("  ... n-contiguous DMA, k-contiguous reads [timing only]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 0, 1, 3>("  ... k-contiguous DMA, tr reads [timing only]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 0, 1, 6>("  ... k-contiguous DMA, the same addresses read with plain ds_read_b64 [timing only]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 0, 1, 4>("  ... k-contiguous DMA, tr reads of [32][16] subtiles [timing only]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 0, 1, 5>("  ... subtile-gather DMA (32 rows x 32 B), k-contiguous reads [timing only]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 32, 1>("sk 256x128 S2 A3 B4 symmetric [traced]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 0>("sk 256x128 S2 A3 B4 ticket hand-off", c, true);
    sk_variant<128, 4, 2, 3, 4, 2, 32>("sk 256x128 S2 A3 B4 ticket [traced]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 128>("sk 256x128 S2 A3 B4 [one K-tile per trip]", c, true);
    sk_variant<128, 4, 2, 3, 3, 2, 0>("sk 256x128 S2 A3 B3", c, true);
    sk_variant<128, 4, 2, 2, 4, 2, 0>("sk 256x128 S2 A2 B4", c, true);
    sk_variant<128, 4, 2, 4, 2, 2, 0>("sk 256x128 S2 A4 B2", c, true);
    sk_variant<128, 4, 2, 3, 4, 2, 4>("sk 256x128 S2 A3 B4 [no exchange]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 4 + 2>("sk 256x128 S2 A3 B4 [no exchange, no DMA]", c, false);
    sk_variant<128, 4, 2, 3, 4, 2, 64>("sk 256x128 S2 A3 B4 [B tile-major]", c, false);
    // round 6: 128 x 128 tiles x split-K 2 = 512 workgroups of 80 KB, TWO per CU (each with its own barrier and hand-off)
    sk_variant<128, 2, 4, 2, 3, 2, 0, 1, 0, 128>("sk 128x128 S2 A2 B3 symmetric, 2 workgroups per CU", c, true);
    sk_variant<128, 2, 4, 3, 2, 2, 0, 1, 0, 128>("sk 128x128 S2 A3 B2 symmetric, 2 workgroups per CU", c, true);
    sk_variant<128, 2, 4, 2, 3, 2, 32, 1, 0, 128>("sk 128x128 S2 A2 B3 symmetric, 2 per CU [traced]", c, false);
    sk_variant<128, 2, 4, 2, 3, 2, 4, 1, 0, 128>("sk 128x128 S2 A2 B3, 2 per CU [no exchange]", c, false);
    sk_variant<128, 2, 4, 2, 3, 2, 4 + 2, 1, 0, 128>("sk 128x128 S2 A2 B3, 2 per CU [no exchange, no DMA]", c, false);
    sk_variant<256, 2, 4, 2, 3, 4, 0>("sk 256x256 S4 A2 B3", c, true);
    sk_variant<256, 2, 4, 2, 3, 4, 32>("sk 256x256 S4 A2 B3 [traced]", c, false);
    sk_variant<256, 2, 4, 2, 3, 4, 4>("sk 256x256 S4 A2 B3 [no exchange]", c, false);
    sk_variant<256, 2, 4, 2, 3, 4, 4 + 2 + 8 + 16>("sk 256x256 S4 [MFMA only]", c, false);
    run_interleaved(c, 9);
    return 0;
}
