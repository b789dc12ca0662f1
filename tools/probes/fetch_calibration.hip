// Known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per ACCESS PATTERN (the guide calibrates
// "FETCH_SIZE = half the bytes" for wide 16-B-per-lane streams only; the config-A kernels load 4-B MFMA fragments, the GEMMs
// use buffer loads and LDS-DMA).  Every kernel touches each byte of a 512 MB buffer (twice the 256 MB memory-side cache)
// exactly once; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o cal -- tools/probes/bin/fetch_calibration
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir> -o cal -- tools/probes/bin/fetch_calibration
// and feed both databases to tools/fetch_calibration.py.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/fetch_calibration.hip -o tools/probes/bin/fetch_calibration
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e__ = (x);                                                              \
        if (e__ != hipSuccess) {                                                           \
            fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e__), __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int64_t BYTES = (int64_t)512 << 20;

// ---- reads: every kernel sums what it loads; the sum is stored only if it hits an impossible value
__global__ __launch_bounds__(256) void cal_read16_stream(const f32x4* in, float* sink, int64_t n) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = in[i];
        s += v[0] + v[1] + v[2] + v[3];
    }
    if (s == 1.2345e30f) sink[0] = s;
}
__global__ __launch_bounds__(256) void cal_read16_stream_nt(const f32x4* in, float* sink, int64_t n) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = __builtin_nontemporal_load(in + i);
        s += v[0] + v[1] + v[2] + v[3];
    }
    if (s == 1.2345e30f) sink[0] = s;
}
__global__ __launch_bounds__(256) void cal_read8_stream(const f32x2* in, float* sink, int64_t n) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x2 v = in[i];
        s += v[0] + v[1];
    }
    if (s == 1.2345e30f) sink[0] = s;
}
__global__ __launch_bounds__(256) void cal_read4_stream(const float* in, float* sink, int64_t n) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += in[i];
    if (s == 1.2345e30f) sink[0] = s;
}
// 4-B loads in the shape of a v_mfma_f32_16x16x4_f32 A fragment: lane l reads row (l & 15), k = 4 step + (l >> 4) of a row-major
// [rows][K] matrix — a wave instruction touches 16 rows x 16 B; a wave walks the whole K of its 16 rows (config A's small-GEMM
// kernels read their operands like this, straight from global memory)
template <int K>
__global__ __launch_bounds__(256) void cal_read4_fragment(const float* in, float* sink, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float s = 0.f;
    for (int64_t panel = wave; panel < rows / 16; panel += n_waves) {
        const float* base = in + (panel * 16 + (lane & 15)) * K + (lane >> 4);
        for (int64_t k = 0; k < K; k += 4) s += base[k];
    }
    if (s == 1.2345e30f) sink[0] = s;
}
// 16-B buffer loads (SGPR resource + VGPR offset), the fp32 GEMM's operand path
__global__ __launch_bounds__(256) void cal_read16_buffer(const f32x4* in, float* sink, int64_t n) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(in), 0, 0xffffffffu, 0x00020000);
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(i * 16), 0, 0);
        s += __uint_as_float(v[0]) + __uint_as_float(v[1]) + __uint_as_float(v[2]) + __uint_as_float(v[3]);
    }
    if (s == 1.2345e30f) sink[0] = s;
}
// LDS-DMA (buffer_load_dwordx4 ... lds), the bf16 GEMMs' operand path: 1 KB per wave instruction into a 16 KB ring per wave
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}
__global__ __launch_bounds__(256) void cal_read16_lds_dma(const char* in, float* sink, int64_t bytes) {
    __shared__ __attribute__((aligned(1024))) char lds[4 * 16384];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, 0xffffffffu, 0x00020000);
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid, n_waves = (int64_t)gridDim.x * 4;
    char* ring = lds + wid * 16384;
    float s = 0.f;
    for (int64_t chunk = wave; chunk < bytes / 16384; chunk += n_waves) {        // 16 KB per wave and trip
#pragma unroll
        for (int j = 0; j < 16; ++j) dma16(rsrc, ring + j * 1024, (uint32_t)(lane * 16), (uint32_t)(chunk * 16384 + j * 1024));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s += *reinterpret_cast<const float*>(ring + lane * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (s == 1.2345e30f) sink[0] = s;
}

// ---- writes
__global__ __launch_bounds__(256) void cal_write16_stream(f32x4* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f32x4{1.f, 2.f, 3.f, (float)i};
}
__global__ __launch_bounds__(256) void cal_write16_stream_nt(f32x4* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, (float)i}, out + i);
}
__global__ __launch_bounds__(256) void cal_write8_stream(f32x2* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f32x2{1.f, (float)i};
}
__global__ __launch_bounds__(256) void cal_write4_stream(float* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (float)i;
}
// read-modify-write in place, 16 B (the optimizer's pattern): BYTES read and BYTES written
__global__ __launch_bounds__(256) void cal_rmw16_stream_nt(f32x4* buf, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = __builtin_nontemporal_load(buf + i);
        v += 1.f;
        __builtin_nontemporal_store(v, buf + i);
    }
}

int main() {
    char *a = nullptr, *b = nullptr;
    float* sink = nullptr;
    CK(hipMalloc(&a, BYTES));
    CK(hipMalloc(&b, BYTES));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(a, 0, BYTES));
    CK(hipMemset(b, 0, BYTES));
    const int grid = 2048;
    // between two measured kernels the OTHER buffer is rewritten, so nothing of the measured one survives in the caches
    auto flush = [&](char* other) { CK(hipMemsetAsync(other, 1, BYTES, 0)); };
    for (int rep = 0; rep < 3; ++rep) {
        flush(b); cal_read16_stream<<<grid, 256>>>((const f32x4*)a, sink, BYTES / 16);
        flush(b); cal_read16_stream_nt<<<grid, 256>>>((const f32x4*)a, sink, BYTES / 16);
        flush(b); cal_read8_stream<<<grid, 256>>>((const f32x2*)a, sink, BYTES / 8);
        flush(b); cal_read4_stream<<<grid, 256>>>((const float*)a, sink, BYTES / 4);
        flush(b); cal_read4_fragment<1024><<<grid, 256>>>((const float*)a, sink, BYTES / 4 / 1024);     // K = 1024 floats
        flush(b); cal_read4_fragment<784><<<grid, 256>>>((const float*)a, sink, BYTES / 4 / 784);       // K = 784 (config A's input width)
        flush(b); cal_read16_buffer<<<grid, 256>>>((const f32x4*)a, sink, BYTES / 16 / 2);              // 32-bit offsets: 256 MB
        flush(b); cal_read16_lds_dma<<<grid, 256>>>(a, sink, BYTES / 2);                                // 256 MB
        flush(b); cal_write16_stream<<<grid, 256>>>((f32x4*)a, BYTES / 16);
        flush(b); cal_write16_stream_nt<<<grid, 256>>>((f32x4*)a, BYTES / 16);
        flush(b); cal_write8_stream<<<grid, 256>>>((f32x2*)a, BYTES / 8);
        flush(b); cal_write4_stream<<<grid, 256>>>((float*)a, BYTES / 4);
        flush(b); cal_rmw16_stream_nt<<<grid, 256>>>((f32x4*)a, BYTES / 16);
    }
    CK(hipDeviceSynchronize());
    // what each kernel moved, for tools/fetch_calibration.py
    const long long full = (long long)BYTES, half = full / 2;
    const long long frag784 = (long long)(BYTES / 4 / 784 / 16) * 16 * 784 * 4;
    printf("KNOWN cal_read16_stream read %lld write 0\n", full);
    printf("KNOWN cal_read16_stream_nt read %lld write 0\n", full);
    printf("KNOWN cal_read8_stream read %lld write 0\n", full);
    printf("KNOWN cal_read4_stream read %lld write 0\n", full);
    printf("KNOWN cal_read4_fragment<1024> read %lld write 0\n", full);
    printf("KNOWN cal_read4_fragment<784> read %lld write 0\n", frag784);
    printf("KNOWN cal_read16_buffer read %lld write 0\n", half);
    printf("KNOWN cal_read16_lds_dma read %lld write 0\n", half);
    printf("KNOWN cal_write16_stream read 0 write %lld\n", full);
    printf("KNOWN cal_write16_stream_nt read 0 write %lld\n", full);
    printf("KNOWN cal_write8_stream read 0 write %lld\n", full);
    printf("KNOWN cal_write4_stream read 0 write %lld\n", full);
    printf("KNOWN cal_rmw16_stream_nt read %lld write %lld\n", full, full);
    return 0;
}
