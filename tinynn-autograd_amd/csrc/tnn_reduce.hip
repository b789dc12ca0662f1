// Reductions (K4): full / row / column sum, max, min and row arg-max.  HBM-bound; wave64 __shfl
// trees + one LDS hop per block; deterministic two-stage schemes instead of atomics so that a step
// replayed from a hipGraph (or run on another rank) reproduces bit-identical gradients.
// Accumulation is in f64 for both f32 and f64 inputs: it is free next to the loads and keeps the
// un-broadcast sums of core/ops.py:41-55 within one fp32 rounding of the reference's float64 result.
#include <math.h>

#include "tnn_internal.h"

namespace {

constexpr int kThreads = 256;
using A = double;

template <int ROP>
__device__ __forceinline__ A r_init() {
    if constexpr (ROP == TNN_RSUM) return 0.0;
    if constexpr (ROP == TNN_RMAX) return -INFINITY;
    return INFINITY;
}
template <int ROP>
__device__ __forceinline__ A r_comb(A a, A b) {
    if constexpr (ROP == TNN_RSUM) return a + b;
    if constexpr (ROP == TNN_RMAX) return b > a ? b : a;
    return b < a ? b : a;
}
template <int ROP>
__device__ __forceinline__ A r_wave(A v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = r_comb<ROP>(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- inner == 1 --------------------------------------------------------------------------------
// (a) short rows: one thread per row
template <typename TI, typename TO, int ROP>
__global__ __launch_bounds__(kThreads) void row_thread_kernel(const TI* __restrict__ in,
                                                              TO* __restrict__ out, int64_t outer,
                                                              int64_t red) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < outer;
         r += (int64_t)gridDim.x * blockDim.x) {
        const TI* p = in + r * red;
        A acc = r_init<ROP>();
        for (int64_t k = 0; k < red; ++k) acc = r_comb<ROP>(acc, (A)p[k]);
        out[r] = (TO)acc;
    }
}
// (b) one wave per row
template <typename TI, typename TO, int ROP>
__global__ __launch_bounds__(kThreads) void row_wave_kernel(const TI* __restrict__ in,
                                                            TO* __restrict__ out, int64_t outer,
                                                            int64_t red) {
    int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < outer; r += nwaves) {
        const TI* p = in + r * red;
        A acc = r_init<ROP>();
        for (int64_t k = lane; k < red; k += 64) acc = r_comb<ROP>(acc, (A)p[k]);
        acc = r_wave<ROP>(acc);
        if (lane == 0) out[r] = (TO)acc;
    }
}
// (c) long rows, few of them: gridDim.x blocks share one row (blockIdx.y = row), one partial each
template <typename TI, typename TO, int ROP>
__global__ __launch_bounds__(kThreads) void row_split_kernel(const TI* __restrict__ in,
                                                             TO* __restrict__ partial,
                                                             int64_t red) {
    __shared__ A lds[kThreads / 64];
    const TI* p = in + (int64_t)blockIdx.y * red;
    A acc = r_init<ROP>();
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < red;
         k += (int64_t)gridDim.x * blockDim.x)
        acc = r_comb<ROP>(acc, (A)p[k]);
    acc = r_wave<ROP>(acc);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        A r = lds[0];
#pragma unroll
        for (int i = 1; i < kThreads / 64; ++i) r = r_comb<ROP>(r, lds[i]);
        partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (TO)r;
    }
}

// ---- inner > 1 : column reduce (the bias gradient, core/ops.py:52-54) --------------------------
// block = 64 columns x 4 row-lanes; blockIdx.x = column strip, blockIdx.y = outer, blockIdx.z =
// slice of `red`.  Loads are coalesced along `inner`.  out is [slice, outer, inner].
template <typename TI, typename TO, int ROP>
__global__ __launch_bounds__(kThreads) void col_kernel(const TI* __restrict__ in,
                                                       TO* __restrict__ out, int64_t red,
                                                       int64_t inner, int64_t rows_per_slice) {
    __shared__ A lds[4][64];
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int64_t c = (int64_t)blockIdx.x * 64 + tx;
    int64_t o = blockIdx.y;
    int64_t r0 = (int64_t)blockIdx.z * rows_per_slice;
    int64_t r1 = r0 + rows_per_slice;
    if (r1 > red) r1 = red;
    A acc = r_init<ROP>();
    if (c < inner) {
        const TI* p = in + (o * red) * inner + c;
        for (int64_t r = r0 + ty; r < r1; r += 4) acc = r_comb<ROP>(acc, (A)p[r * inner]);
    }
    lds[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < inner) {
        A v = lds[0][tx];
        v = r_comb<ROP>(v, lds[1][tx]);
        v = r_comb<ROP>(v, lds[2][tx]);
        v = r_comb<ROP>(v, lds[3][tx]);
        out[((int64_t)blockIdx.z * gridDim.y + o) * inner + c] = (TO)v;
    }
}

template <typename T, int ROP>
int reduce_typed(const void* in, void* out, int64_t outer, int64_t red, int64_t inner) {
    hipStream_t s = tnn::stream();
    const int64_t block_cap = (int64_t)tnn::num_cus() * 8;
    if (inner == 1) {
        if (red < 64) {
            hipLaunchKernelGGL((row_thread_kernel<T, T, ROP>), tnn::stream_grid(outer, kThreads),
                               kThreads, 0, s, (const T*)in, (T*)out, outer, red);
            TNN_LAUNCH_OK();
            return 0;
        }
        if (outer >= 64 || red < 8192) {
            hipLaunchKernelGGL((row_wave_kernel<T, T, ROP>), tnn::stream_grid(outer * 64, kThreads),
                               kThreads, 0, s, (const T*)in, (T*)out, outer, red);
            TNN_LAUNCH_OK();
            return 0;
        }
        // few long rows: split each over nb blocks, then combine the f64 partials
        int64_t nb = (red + (int64_t)kThreads * 16 - 1) / ((int64_t)kThreads * 16);
        int64_t cap = block_cap / outer;
        if (cap < 1) cap = 1;
        if (nb > cap) nb = cap;
        if (nb > 1024) nb = 1024;
        void* ws = nullptr;
        if (tnn_malloc((size_t)(outer * nb) * sizeof(A), &ws)) return 1;
        hipLaunchKernelGGL((row_split_kernel<T, A, ROP>), dim3((unsigned)nb, (unsigned)outer),
                           kThreads, 0, s, (const T*)in, (A*)ws, red);
        hipLaunchKernelGGL((row_wave_kernel<A, T, ROP>), tnn::stream_grid(outer * 64, kThreads),
                           kThreads, 0, s, (const A*)ws, (T*)out, outer, nb);
        tnn_free(ws);   // stream-ordered: the pool may hand it out again only to later work
        TNN_LAUNCH_OK();
        return 0;
    }
    // column reduce
    int64_t strips = (inner + 63) / 64;
    int64_t slices = 1;
    if (strips * outer < block_cap / 4 && red >= 256) {
        slices = (block_cap / 2) / (strips * outer);
        int64_t max_slices = red / 64;          // >= 64 rows per slice
        if (slices > max_slices) slices = max_slices;
        if (slices > 256) slices = 256;
        if (slices < 1) slices = 1;
    }
    int64_t rows_per_slice = (red + slices - 1) / slices;
    slices = (red + rows_per_slice - 1) / rows_per_slice;
    dim3 grid((unsigned)strips, (unsigned)outer, (unsigned)slices);
    if (slices == 1) {
        hipLaunchKernelGGL((col_kernel<T, T, ROP>), grid, kThreads, 0, s, (const T*)in, (T*)out, red,
                           inner, rows_per_slice);
        TNN_LAUNCH_OK();
        return 0;
    }
    void* ws = nullptr;
    int64_t oi = outer * inner;
    if (tnn_malloc((size_t)(slices * oi) * sizeof(A), &ws)) return 1;
    hipLaunchKernelGGL((col_kernel<T, A, ROP>), grid, kThreads, 0, s, (const T*)in, (A*)ws, red, inner,
                       rows_per_slice);
    // second pass: ws is [slices, outer*inner] -> reduce over slices
    hipLaunchKernelGGL((col_kernel<A, T, ROP>), dim3((unsigned)((oi + 63) / 64), 1, 1), kThreads, 0, s,
                       (const A*)ws, (T*)out, slices, oi, slices);
    tnn_free(ws);
    TNN_LAUNCH_OK();
    return 0;
}

template <typename T>
int reduce_dispatch(int rop, const void* in, void* out, int64_t outer, int64_t red, int64_t inner) {
    switch (rop) {
        case TNN_RSUM: return reduce_typed<T, TNN_RSUM>(in, out, outer, red, inner);
        case TNN_RMAX: return reduce_typed<T, TNN_RMAX>(in, out, outer, red, inner);
        case TNN_RMIN: return reduce_typed<T, TNN_RMIN>(in, out, outer, red, inner);
    }
    tnn::set_error("tnn_reduce: unknown reduction %d", rop);
    return 2;
}

// first index of the row maximum (numpy argmax tie rule): one wave per row
template <typename T>
__global__ __launch_bounds__(kThreads) void argmax_rows_kernel(const T* __restrict__ in,
                                                               int64_t* __restrict__ out,
                                                               int64_t rows, int64_t cols) {
    int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const T* p = in + r * cols;
        T best = -INFINITY;
        int64_t bi = INT64_MAX;
        for (int64_t k = lane; k < cols; k += 64) {
            T v = p[k];
            if (v > best || bi == INT64_MAX) { best = v; bi = k; }   // strictly greater keeps the first
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            T ov = __shfl_xor(best, o, 64);
            int64_t oi = __shfl_xor(bi, o, 64);
            if (oi != INT64_MAX && (bi == INT64_MAX || ov > best || (ov == best && oi < bi))) {
                best = ov;
                bi = oi;
            }
        }
        if (lane == 0) out[r] = bi == INT64_MAX ? 0 : bi;
    }
}

}  // namespace

extern "C" {

int tnn_reduce(int rop, const void* in, void* out, int64_t outer, int64_t red, int64_t inner,
               int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(outer >= 0 && red >= 0 && inner >= 0, "tnn_reduce: negative extent");
    if (outer * inner == 0) return 0;
    TNN_REQUIRE(red > 0 || rop == TNN_RSUM, "tnn_reduce: max/min of an empty axis");
    TNN_REQUIRE(outer <= 65535 || inner == 1, "tnn_reduce: outer %lld too large", (long long)outer);
    switch (dtype) {
        case TNN_F32: return reduce_dispatch<float>(rop, in, out, outer, red, inner);
        case TNN_F64: return reduce_dispatch<double>(rop, in, out, outer, red, inner);
    }
    tnn::set_error("tnn_reduce: dtype %d is not a float type", dtype);
    return 2;
}

int tnn_argmax_rows(const void* in, void* out_i64, int64_t rows, int64_t cols, int dtype) {
    TNN_NEED_INIT();
    if (rows <= 0) return 0;
    TNN_REQUIRE(cols > 0, "tnn_argmax_rows: empty rows");
    unsigned grid = tnn::stream_grid(rows * 64, kThreads);
    switch (dtype) {
        case TNN_F32:
            hipLaunchKernelGGL((argmax_rows_kernel<float>), grid, kThreads, 0, tnn::stream(),
                               (const float*)in, (int64_t*)out_i64, rows, cols);
            break;
        case TNN_F64:
            hipLaunchKernelGGL((argmax_rows_kernel<double>), grid, kThreads, 0, tnn::stream(),
                               (const double*)in, (int64_t*)out_i64, rows, cols);
            break;
        default: tnn::set_error("tnn_argmax_rows: dtype %d is not a float type", dtype); return 2;
    }
    TNN_LAUNCH_OK();
    return 0;
}

}  // extern "C"
