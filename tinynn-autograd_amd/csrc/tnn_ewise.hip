// Elementwise, broadcast, compare, cast and strided-copy kernels (K2, K3, K5, K6, K7 of SURVEY §8a).
// All HBM-bound: 16 B per lane per access where the layout allows it, grid capped at 8 blocks/CU
// with a grid-stride loop, no LDS.  Broadcasting is expressed as element strides (0 = broadcast)
// after the host collapses adjacent dimensions, so the common cases ([m,n] op [1,n], same-shape,
// scalar) run without integer division.
#include <math.h>

#include "tnn_internal.h"

namespace {

constexpr int kMaxDim = 6;
constexpr int kThreads = 256;

struct Dims {
    int nd;
    int64_t shape[kMaxDim];
    int64_t sa[kMaxDim];
    int64_t sb[kMaxDim];
};

// ---------------------------------------------------------------- functors
template <typename T> __device__ __forceinline__ T t_pow(T a, T b);
template <> __device__ __forceinline__ float t_pow<float>(float a, float b) { return powf(a, b); }
template <> __device__ __forceinline__ double t_pow<double>(double a, double b) { return pow(a, b); }
template <typename T> __device__ __forceinline__ T t_exp(T a);
template <> __device__ __forceinline__ float t_exp<float>(float a) { return expf(a); }
template <> __device__ __forceinline__ double t_exp<double>(double a) { return exp(a); }
template <typename T> __device__ __forceinline__ T t_log(T a);
template <> __device__ __forceinline__ float t_log<float>(float a) { return logf(a); }
template <> __device__ __forceinline__ double t_log<double>(double a) { return log(a); }
template <typename T> __device__ __forceinline__ T t_sqrt(T a);
template <> __device__ __forceinline__ float t_sqrt<float>(float a) { return sqrtf(a); }
template <> __device__ __forceinline__ double t_sqrt<double>(double a) { return sqrt(a); }
template <typename T> __device__ __forceinline__ T t_tanh(T a);
template <> __device__ __forceinline__ float t_tanh<float>(float a) { return tanhf(a); }
template <> __device__ __forceinline__ double t_tanh<double>(double a) { return tanh(a); }

template <typename T, int OP>
__device__ __forceinline__ T bin(T a, T b) {
    if constexpr (OP == TNN_ADD) return a + b;
    if constexpr (OP == TNN_SUB) return a - b;
    if constexpr (OP == TNN_MUL) return a * b;
    if constexpr (OP == TNN_DIV) return a / b;
    if constexpr (OP == TNN_POW) return t_pow<T>(a, b);
    if constexpr (OP == TNN_MAX) return a >= b ? a : b;   // np.maximum; ties irrelevant for values
    if constexpr (OP == TNN_MIN) return a <= b ? a : b;
    return a;
}

template <typename T, int CMP>
__device__ __forceinline__ uint8_t cmp(T a, T b) {
    if constexpr (CMP == TNN_GT) return a > b;
    if constexpr (CMP == TNN_GE) return a >= b;
    if constexpr (CMP == TNN_LT) return a < b;
    if constexpr (CMP == TNN_LE) return a <= b;
    if constexpr (CMP == TNN_EQ) return a == b;
    if constexpr (CMP == TNN_NE) return a != b;
    return 0;
}

template <typename T, int OP>
__device__ __forceinline__ T una(T a) {
    if constexpr (OP == TNN_NEG) return -a;
    if constexpr (OP == TNN_EXP) return t_exp<T>(a);
    if constexpr (OP == TNN_LOG) return t_log<T>(a);
    if constexpr (OP == TNN_SQRT) return t_sqrt<T>(a);
    if constexpr (OP == TNN_SQUARE) return a * a;
    if constexpr (OP == TNN_ABS) return a < T(0) ? -a : a;
    if constexpr (OP == TNN_RECIP) return T(1) / a;
    if constexpr (OP == TNN_SIGMOID) return T(1) / (T(1) + t_exp<T>(-a));
    if constexpr (OP == TNN_TANH) return t_tanh<T>(a);
    return a;
}

// 16-byte vector of T
template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

template <typename T>
__device__ __forceinline__ void vload(const T* p, T (&r)[Vec16<T>::N]) {
    using V = typename Vec16<T>::type;
    V v = *reinterpret_cast<const V*>(p);
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) r[i] = e[i];
}
template <typename T>
__device__ __forceinline__ void vstore(T* p, const T (&r)[Vec16<T>::N]) {
    using V = typename Vec16<T>::type;
    V v;
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) e[i] = r[i];
    *reinterpret_cast<V*>(p) = v;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------- flat kernels (n elements)
// F(i-th element values...) applied to dense arrays; VEC path when all pointers are 16-B aligned.
template <typename T, bool VEC, typename F>
__global__ __launch_bounds__(kThreads) void flat2_kernel(const T* __restrict__ a,
                                                         const T* __restrict__ b,
                                                         T* __restrict__ out, int64_t n, F f) {
    constexpr int N = Vec16<T>::N;
    int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nth = (int64_t)gridDim.x * blockDim.x;
    if constexpr (VEC) {
        int64_t nv = n / N;
        for (int64_t i = tid; i < nv; i += nth) {
            T x[N], y[N], r[N];
            vload<T>(a + i * N, x);
            vload<T>(b + i * N, y);
#pragma unroll
            for (int k = 0; k < N; ++k) r[k] = f(x[k], y[k]);
            vstore<T>(out + i * N, r);
        }
        for (int64_t i = nv * N + tid; i < n; i += nth) out[i] = f(a[i], b[i]);
    } else {
        for (int64_t i = tid; i < n; i += nth) out[i] = f(a[i], b[i]);
    }
}

template <typename T, bool VEC, typename F>
__global__ __launch_bounds__(kThreads) void flat1_kernel(const T* __restrict__ a,
                                                         T* __restrict__ out, int64_t n, F f) {
    constexpr int N = Vec16<T>::N;
    int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nth = (int64_t)gridDim.x * blockDim.x;
    if constexpr (VEC) {
        int64_t nv = n / N;
        for (int64_t i = tid; i < nv; i += nth) {
            T x[N], r[N];
            vload<T>(a + i * N, x);
#pragma unroll
            for (int k = 0; k < N; ++k) r[k] = f(x[k]);
            vstore<T>(out + i * N, r);
        }
        for (int64_t i = nv * N + tid; i < n; i += nth) out[i] = f(a[i]);
    } else {
        for (int64_t i = tid; i < n; i += nth) out[i] = f(a[i]);
    }
}

template <typename T, typename F>
int launch_flat2(const void* a, const void* b, void* out, int64_t n, F f) {
    if (n <= 0) return 0;
    bool vec = aligned16(a) && aligned16(b) && aligned16(out);
    int64_t items = vec ? (n + Vec16<T>::N - 1) / Vec16<T>::N : n;
    unsigned grid = tnn::stream_grid(items, kThreads);
    if (vec)
        hipLaunchKernelGGL((flat2_kernel<T, true, F>), grid, kThreads, 0, tnn::stream(),
                           (const T*)a, (const T*)b, (T*)out, n, f);
    else
        hipLaunchKernelGGL((flat2_kernel<T, false, F>), grid, kThreads, 0, tnn::stream(),
                           (const T*)a, (const T*)b, (T*)out, n, f);
    TNN_LAUNCH_OK();
    return 0;
}

template <typename T, typename F>
int launch_flat1(const void* a, void* out, int64_t n, F f) {
    if (n <= 0) return 0;
    bool vec = aligned16(a) && aligned16(out);
    int64_t items = vec ? (n + Vec16<T>::N - 1) / Vec16<T>::N : n;
    unsigned grid = tnn::stream_grid(items, kThreads);
    if (vec)
        hipLaunchKernelGGL((flat1_kernel<T, true, F>), grid, kThreads, 0, tnn::stream(),
                           (const T*)a, (T*)out, n, f);
    else
        hipLaunchKernelGGL((flat1_kernel<T, false, F>), grid, kThreads, 0, tnn::stream(),
                           (const T*)a, (T*)out, n, f);
    TNN_LAUNCH_OK();
    return 0;
}

// ---------------------------------------------------------------- strided (broadcast) kernels
// TO = output element type (T for arithmetic, uint8_t for comparisons).
// 2-D fast path: threads run along the inner dimension (coalesced), blocks walk rows -> no division.
template <typename T, typename TO, typename F>
__global__ __launch_bounds__(kThreads) void strided2d_kernel(const T* __restrict__ a,
                                                             const T* __restrict__ b,
                                                             TO* __restrict__ out, int64_t R,
                                                             int64_t C, int64_t sa0, int64_t sa1,
                                                             int64_t sb0, int64_t sb1, F f) {
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y) {
        const T* ar = a + r * sa0;
        const T* br = b + r * sb0;
        TO* orow = out + r * C;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < C;
             c += (int64_t)gridDim.x * blockDim.x)
            orow[c] = f(ar[c * sa1], br[c * sb1]);
    }
}

// generic N-d: decompose the flat output index (only used for >2 collapsed dims)
template <typename T, typename TO, typename F>
__global__ __launch_bounds__(kThreads) void stridednd_kernel(const T* __restrict__ a,
                                                             const T* __restrict__ b,
                                                             TO* __restrict__ out, int64_t n,
                                                             Dims d, F f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = i, oa = 0, ob = 0;
#pragma unroll
        for (int k = kMaxDim - 1; k >= 0; --k) {
            if (k < d.nd) {
                int64_t q = rem / d.shape[k];
                int64_t idx = rem - q * d.shape[k];
                rem = q;
                oa += idx * d.sa[k];
                ob += idx * d.sb[k];
            }
        }
        out[i] = f(a[oa], b[ob]);
    }
}

// collapse adjacent dims that both operands walk contiguously; drop size-1 dims
Dims collapse(int ndim, const int64_t* shape, const int64_t* sa, const int64_t* sb) {
    Dims d;
    d.nd = 0;
    for (int k = 0; k < ndim; ++k) {
        if (shape[k] == 1) continue;
        int64_t a = sa ? sa[k] : 0, b = sb ? sb[k] : 0;
        if (d.nd > 0) {
            int j = d.nd - 1;
            if (d.sa[j] == a * shape[k] && d.sb[j] == b * shape[k]) {   // mergeable
                d.shape[j] *= shape[k];
                d.sa[j] = a;
                d.sb[j] = b;
                continue;
            }
        }
        d.shape[d.nd] = shape[k];
        d.sa[d.nd] = a;
        d.sb[d.nd] = b;
        d.nd++;
    }
    if (d.nd == 0) {
        d.nd = 1;
        d.shape[0] = 1;
        d.sa[0] = 0;
        d.sb[0] = 0;
    }
    for (int k = d.nd; k < kMaxDim; ++k) { d.shape[k] = 1; d.sa[k] = 0; d.sb[k] = 0; }
    return d;
}

template <typename T, typename TO, typename F>
int launch_strided(const void* a, const void* b, void* out, const Dims& d, F f) {
    int64_t n = 1;
    for (int k = 0; k < d.nd; ++k) n *= d.shape[k];
    if (n <= 0) return 0;
    if (d.nd <= 2) {
        int64_t R = d.nd == 2 ? d.shape[0] : 1, C = d.nd == 2 ? d.shape[1] : d.shape[0];
        int64_t sa0 = d.nd == 2 ? d.sa[0] : 0, sa1 = d.nd == 2 ? d.sa[1] : d.sa[0];
        int64_t sb0 = d.nd == 2 ? d.sb[0] : 0, sb1 = d.nd == 2 ? d.sb[1] : d.sb[0];
        unsigned gx = (unsigned)((C + kThreads - 1) / kThreads);
        int64_t cap = (int64_t)tnn::num_cus() * 8;
        if (gx > cap) gx = (unsigned)cap;
        int64_t gy = cap / gx;
        if (gy > R) gy = R;
        if (gy < 1) gy = 1;
        if (gy > 65535) gy = 65535;
        hipLaunchKernelGGL((strided2d_kernel<T, TO, F>), dim3(gx, (unsigned)gy), kThreads, 0,
                           tnn::stream(), (const T*)a, (const T*)b, (TO*)out, R, C, sa0, sa1, sb0,
                           sb1, f);
    } else {
        hipLaunchKernelGGL((stridednd_kernel<T, TO, F>), tnn::stream_grid(n, kThreads), kThreads, 0,
                           tnn::stream(), (const T*)a, (const T*)b, (TO*)out, n, d, f);
    }
    TNN_LAUNCH_OK();
    return 0;
}

bool is_dense(const Dims& d, const int64_t* s) {
    int64_t expect = 1;
    for (int k = d.nd - 1; k >= 0; --k) {
        if (s[k] != expect) return false;
        expect *= d.shape[k];
    }
    return true;
}

template <typename T, int OP>
struct BinF {
    __device__ __forceinline__ T operator()(T a, T b) const { return bin<T, OP>(a, b); }
};
template <typename T, int CMP>
struct CmpF {
    __device__ __forceinline__ uint8_t operator()(T a, T b) const { return cmp<T, CMP>(a, b); }
};
template <typename T, int OP>
struct UnaF {
    __device__ __forceinline__ T operator()(T a) const { return una<T, OP>(a); }
};
template <typename T, int OP, bool LHS>
struct ScalarF {
    T s;
    __device__ __forceinline__ T operator()(T a) const {
        return LHS ? bin<T, OP>(s, a) : bin<T, OP>(a, s);
    }
};
template <typename T, int CMP>
struct CmpScalarF {
    T s;
    __device__ __forceinline__ uint8_t operator()(T a) const { return cmp<T, CMP>(a, s); }
};

template <typename T, int OP>
int binary_typed(const void* a, const void* b, void* out, const Dims& d) {
    if (is_dense(d, d.sa) && is_dense(d, d.sb)) {
        int64_t n = 1;
        for (int k = 0; k < d.nd; ++k) n *= d.shape[k];
        return launch_flat2<T>(a, b, out, n, BinF<T, OP>{});
    }
    return launch_strided<T, T>(a, b, out, d, BinF<T, OP>{});
}

template <typename T>
int binary_dispatch(int op, const void* a, const void* b, void* out, const Dims& d) {
    switch (op) {
        case TNN_ADD: return binary_typed<T, TNN_ADD>(a, b, out, d);
        case TNN_SUB: return binary_typed<T, TNN_SUB>(a, b, out, d);
        case TNN_MUL: return binary_typed<T, TNN_MUL>(a, b, out, d);
        case TNN_DIV: return binary_typed<T, TNN_DIV>(a, b, out, d);
        case TNN_POW: return binary_typed<T, TNN_POW>(a, b, out, d);
        case TNN_MAX: return binary_typed<T, TNN_MAX>(a, b, out, d);
        case TNN_MIN: return binary_typed<T, TNN_MIN>(a, b, out, d);
    }
    tnn::set_error("tnn_ewise_binary: unknown op %d", op);
    return 2;
}

template <typename T>
int compare_dispatch(int c, const void* a, const void* b, void* out, const Dims& d) {
    switch (c) {
        case TNN_GT: return launch_strided<T, uint8_t>(a, b, out, d, CmpF<T, TNN_GT>{});
        case TNN_GE: return launch_strided<T, uint8_t>(a, b, out, d, CmpF<T, TNN_GE>{});
        case TNN_LT: return launch_strided<T, uint8_t>(a, b, out, d, CmpF<T, TNN_LT>{});
        case TNN_LE: return launch_strided<T, uint8_t>(a, b, out, d, CmpF<T, TNN_LE>{});
        case TNN_EQ: return launch_strided<T, uint8_t>(a, b, out, d, CmpF<T, TNN_EQ>{});
        case TNN_NE: return launch_strided<T, uint8_t>(a, b, out, d, CmpF<T, TNN_NE>{});
    }
    tnn::set_error("tnn_ewise_compare: unknown comparison %d", c);
    return 2;
}

template <typename T, bool LHS>
int scalar_dispatch(int op, const void* a, double s, void* out, int64_t n) {
    T sv = (T)s;
    switch (op) {
        case TNN_ADD: return launch_flat1<T>(a, out, n, ScalarF<T, TNN_ADD, LHS>{sv});
        case TNN_SUB: return launch_flat1<T>(a, out, n, ScalarF<T, TNN_SUB, LHS>{sv});
        case TNN_MUL: return launch_flat1<T>(a, out, n, ScalarF<T, TNN_MUL, LHS>{sv});
        case TNN_DIV: return launch_flat1<T>(a, out, n, ScalarF<T, TNN_DIV, LHS>{sv});
        case TNN_POW:
            if (!LHS && s == 2.0) return launch_flat1<T>(a, out, n, UnaF<T, TNN_SQUARE>{});
            if (!LHS && s == 0.5) return launch_flat1<T>(a, out, n, UnaF<T, TNN_SQRT>{});
            if (!LHS && s == 1.0) return launch_flat1<T>(a, out, n, UnaF<T, TNN_COPY>{});
            return launch_flat1<T>(a, out, n, ScalarF<T, TNN_POW, LHS>{sv});
        case TNN_MAX: return launch_flat1<T>(a, out, n, ScalarF<T, TNN_MAX, LHS>{sv});
        case TNN_MIN: return launch_flat1<T>(a, out, n, ScalarF<T, TNN_MIN, LHS>{sv});
    }
    tnn::set_error("tnn_ewise_scalar: unknown op %d", op);
    return 2;
}

template <typename T>
int unary_dispatch(int op, const void* in, void* out, int64_t n) {
    switch (op) {
        case TNN_NEG: return launch_flat1<T>(in, out, n, UnaF<T, TNN_NEG>{});
        case TNN_EXP: return launch_flat1<T>(in, out, n, UnaF<T, TNN_EXP>{});
        case TNN_LOG: return launch_flat1<T>(in, out, n, UnaF<T, TNN_LOG>{});
        case TNN_SQRT: return launch_flat1<T>(in, out, n, UnaF<T, TNN_SQRT>{});
        case TNN_SQUARE: return launch_flat1<T>(in, out, n, UnaF<T, TNN_SQUARE>{});
        case TNN_ABS: return launch_flat1<T>(in, out, n, UnaF<T, TNN_ABS>{});
        case TNN_RECIP: return launch_flat1<T>(in, out, n, UnaF<T, TNN_RECIP>{});
        case TNN_SIGMOID: return launch_flat1<T>(in, out, n, UnaF<T, TNN_SIGMOID>{});
        case TNN_TANH: return launch_flat1<T>(in, out, n, UnaF<T, TNN_TANH>{});
        case TNN_COPY: return launch_flat1<T>(in, out, n, UnaF<T, TNN_COPY>{});
    }
    tnn::set_error("tnn_ewise_unary: unknown op %d", op);
    return 2;
}

template <typename T>
struct ClipF {
    T lo, hi;
    bool has_lo, has_hi;
    __device__ __forceinline__ T operator()(T a) const {
        if (has_lo && a < lo) a = lo;
        if (has_hi && a > hi) a = hi;
        return a;
    }
};
template <typename T>
struct ClipBwdF {
    T lo, hi;
    bool has_lo, has_hi;
    __device__ __forceinline__ T operator()(T g, T x) const {
        bool keep = (!has_lo || x >= lo) && (!has_hi || x <= hi);
        return keep ? g : T(0);
    }
};
template <typename T>
struct SignMaskF {      // g where the sign bit of y is clear, else 0 (y = a sign-encoded ReLU output: -0.0 marks z < 0)
    __device__ __forceinline__ T operator()(T g, T y) const { return signbit(y) ? T(0) : g; }
};
template <typename T>
struct AxpyF {
    T alpha;
    __device__ __forceinline__ T operator()(T y, T x) const { return y + alpha * x; }
};

// one-input kernels whose output type differs from the input type
template <typename TI, typename TO, typename F>
__global__ __launch_bounds__(kThreads) void map_kernel(const TI* __restrict__ in,
                                                       TO* __restrict__ out, int64_t n, F f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f(in[i]);
}
template <typename TI, typename TO>
struct CastF {
    __device__ __forceinline__ TO operator()(TI a) const { return (TO)a; }
};
template <typename TI>
struct ToBoolF {
    __device__ __forceinline__ uint8_t operator()(TI a) const { return a != TI(0); }
};

template <typename TI, typename TO, typename F>
int launch_map(const void* in, void* out, int64_t n, F f) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL((map_kernel<TI, TO, F>), tnn::stream_grid(n, kThreads), kThreads, 0,
                       tnn::stream(), (const TI*)in, (TO*)out, n, f);
    TNN_LAUNCH_OK();
    return 0;
}

template <typename TI>
int cast_from(const void* in, void* out, int out_dtype, int64_t n) {
    switch (out_dtype) {
        case TNN_F32: return launch_map<TI, float>(in, out, n, CastF<TI, float>{});
        case TNN_F64: return launch_map<TI, double>(in, out, n, CastF<TI, double>{});
        case TNN_I64: return launch_map<TI, int64_t>(in, out, n, CastF<TI, int64_t>{});
        case TNN_U8: return launch_map<TI, uint8_t>(in, out, n, ToBoolF<TI>{});
    }
    tnn::set_error("tnn_cast: unknown output dtype %d", out_dtype);
    return 2;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void mul_mask_kernel(const T* __restrict__ g,
                                                            const uint8_t* __restrict__ mask,
                                                            T* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = mask[i] ? g[i] : T(0);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void fill_kernel(T* __restrict__ out, int64_t n, T v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = v;
}

// gather (strided read -> dense write) and scatter (dense read -> strided write)
struct Dims1 {
    int nd;
    int64_t shape[kMaxDim];
    int64_t st[kMaxDim];
};
template <typename T, bool SCATTER>
__global__ __launch_bounds__(kThreads) void strided_move_kernel(const T* __restrict__ in,
                                                                T* __restrict__ out, int64_t n,
                                                                Dims1 d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = i, off = 0;
#pragma unroll
        for (int k = kMaxDim - 1; k >= 0; --k) {
            if (k < d.nd) {
                int64_t q = rem / d.shape[k];
                off += (rem - q * d.shape[k]) * d.st[k];
                rem = q;
            }
        }
        if (SCATTER) out[off] = in[i];
        else out[i] = in[off];
    }
}

// 2-D transpose through LDS (the only strided copy that is bandwidth-critical): 64x64 tile,
// +1 padding -> conflict-free column reads.
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* __restrict__ in,
                                                          T* __restrict__ out, int64_t R,
                                                          int64_t C) {
    __shared__ T tile[64][65];
    int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int j = ty; j < 64; j += 4) {
        int64_t r = r0 + j, c = c0 + tx;
        if (r < R && c < C) tile[j][tx] = in[r * C + c];
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        int64_t c = c0 + j, r = r0 + tx;   // out is [C, R]
        if (c < C && r < R) out[c * R + r] = tile[tx][j];
    }
}

template <typename T>
int strided_move(const void* in, void* out, const int64_t* st, int ndim, const int64_t* shape,
                 bool scatter) {
    Dims1 d;
    d.nd = 0;
    int64_t n = 1;
    for (int k = 0; k < ndim; ++k) {
        n *= shape[k];
        if (shape[k] == 1) continue;
        if (d.nd > 0 && d.st[d.nd - 1] == st[k] * shape[k]) {
            d.shape[d.nd - 1] *= shape[k];
            d.st[d.nd - 1] = st[k];
            continue;
        }
        d.shape[d.nd] = shape[k];
        d.st[d.nd] = st[k];
        d.nd++;
    }
    if (n <= 0) return 0;
    if (d.nd == 0) { d.nd = 1; d.shape[0] = 1; d.st[0] = 0; }
    for (int k = d.nd; k < kMaxDim; ++k) { d.shape[k] = 1; d.st[k] = 0; }
    if (!scatter && d.nd == 2 && d.st[0] == 1 && d.st[1] == d.shape[0]) {
        // dense 2-D transpose: out[i,j] = in[j,i]; in is [shape1, shape0]
        int64_t R = d.shape[1], C = d.shape[0];
        dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
        if (grid.y <= 65535) {
            hipLaunchKernelGGL((transpose2d_kernel<T>), grid, 256, 0, tnn::stream(), (const T*)in,
                               (T*)out, R, C);
            TNN_LAUNCH_OK();
            return 0;
        }
    }
    if (scatter)
        hipLaunchKernelGGL((strided_move_kernel<T, true>), tnn::stream_grid(n, kThreads), kThreads,
                           0, tnn::stream(), (const T*)in, (T*)out, n, d);
    else
        hipLaunchKernelGGL((strided_move_kernel<T, false>), tnn::stream_grid(n, kThreads), kThreads,
                           0, tnn::stream(), (const T*)in, (T*)out, n, d);
    TNN_LAUNCH_OK();
    return 0;
}

// rows: one wave-sized chunk of a row per thread group; row_elems small (784) .. large
template <typename T, bool SCATTER>
__global__ __launch_bounds__(kThreads) void rows_kernel(const T* __restrict__ src,
                                                        const int64_t* __restrict__ idx,
                                                        T* __restrict__ dst, int64_t n_idx,
                                                        int64_t row_elems, int64_t limit) {
    for (int64_t r = blockIdx.y; r < n_idx; r += gridDim.y) {
        int64_t j = idx[r];
        if (j < 0) j += limit;            // numpy negative indices
        if (j < 0 || j >= limit) continue;   // bounds are validated on the host; stay memory-safe
        const T* s = SCATTER ? src + r * row_elems : src + j * row_elems;
        T* d = SCATTER ? dst + j * row_elems : dst + r * row_elems;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < row_elems;
             c += (int64_t)gridDim.x * blockDim.x)
            d[c] = s[c];
    }
}

// The epoch gather (utils/data_iterator.py:26-28: inputs[idx] of the whole 50,000 x 784 training set, 157 MB): rows whose byte
// length and both bases are multiples of 16 move as 16-B pieces, several rows per workgroup (a 784-float row is 196 pieces);
// the source is read with streaming loads — it will not be read again this epoch, the gathered copy will, batch by batch.
typedef float f32x4_rows __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kThreads) void gather_rows16_kernel(const f32x4_rows* __restrict__ src, const int64_t* __restrict__ idx,
                                                                 f32x4_rows* __restrict__ dst, int64_t n_idx, int64_t row_v, int64_t limit) {
    const int64_t total = n_idx * row_v;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / row_v, c = i - r * row_v;
        int64_t j = idx[r];
        if (j < 0) j += limit;
        if (j < 0 || j >= limit) continue;
        dst[i] = __builtin_nontemporal_load(src + j * row_v + c);
    }
}

template <typename T>
int rows_move(const void* src, const void* idx, void* dst, int64_t n_idx, int64_t row_elems,
              int64_t limit, bool scatter) {
    if (n_idx <= 0 || row_elems <= 0) return 0;
    const int64_t row_bytes = row_elems * (int64_t)sizeof(T);
    if (!scatter && row_bytes % 16 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const int64_t row_v = row_bytes / 16;
        hipLaunchKernelGGL(gather_rows16_kernel, tnn::stream_grid(n_idx * row_v, kThreads), kThreads, 0, tnn::stream(),
                           (const f32x4_rows*)src, (const int64_t*)idx, (f32x4_rows*)dst, n_idx, row_v, limit);
        TNN_LAUNCH_OK();
        return 0;
    }
    unsigned gx = (unsigned)((row_elems + kThreads - 1) / kThreads);
    if (gx > 64) gx = 64;
    int64_t gy = n_idx;
    if (gy > 65535) gy = 65535;
    if (scatter)
        hipLaunchKernelGGL((rows_kernel<T, true>), dim3(gx, (unsigned)gy), kThreads, 0, tnn::stream(),
                           (const T*)src, (const int64_t*)idx, (T*)dst, n_idx, row_elems, limit);
    else
        hipLaunchKernelGGL((rows_kernel<T, false>), dim3(gx, (unsigned)gy), kThreads, 0,
                           tnn::stream(), (const T*)src, (const int64_t*)idx, (T*)dst, n_idx,
                           row_elems, limit);
    TNN_LAUNCH_OK();
    return 0;
}

// out[i] = *ptrs[i]: n scalars that live in n separate device buffers (a training loop's per-step 0-d losses, core/tensor.py
// values of examples/mnist/run.py:84's loss_list) into one vector — ONE launch instead of n 4-byte device copies
template <typename T>
__global__ __launch_bounds__(kThreads) void gather_scalars_kernel(const uint64_t* __restrict__ ptrs, T* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = *reinterpret_cast<const T*>(ptrs[i]);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void one_hot_kernel(const int64_t* __restrict__ labels,
                                                           T* __restrict__ out, int64_t n,
                                                           int64_t classes) {
    int64_t total = n * classes;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / classes, c = i - r * classes;
        out[i] = labels[r] == c ? T(1) : T(0);
    }
}

size_t dtype_size(int dtype) {
    switch (dtype) {
        case TNN_F32: return 4;
        case TNN_F64: return 8;
        case TNN_I64: return 8;
        case TNN_U8: return 1;
    }
    return 0;
}

}  // namespace

#define TNN_FLOAT_SWITCH(dtype, fn_name, CALL)                                           \
    switch (dtype) {                                                                     \
        case TNN_F32: { using T = float; return CALL; }                                  \
        case TNN_F64: { using T = double; return CALL; }                                 \
        default: tnn::set_error(fn_name ": dtype %d is not a float type", dtype); return 2; \
    }

extern "C" {

int tnn_ewise_binary(int op, const void* a, const int64_t* stride_a, const void* b,
                     const int64_t* stride_b, void* out, int ndim, const int64_t* shape,
                     int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(ndim >= 0 && ndim <= kMaxDim, "tnn_ewise_binary: ndim %d > %d", ndim, kMaxDim);
    Dims d = collapse(ndim, shape, stride_a, stride_b);
    TNN_FLOAT_SWITCH(dtype, "tnn_ewise_binary", (binary_dispatch<T>(op, a, b, out, d)));
}

int tnn_ewise_compare(int c, const void* a, const int64_t* stride_a, const void* b,
                      const int64_t* stride_b, void* out_u8, int ndim, const int64_t* shape,
                      int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(ndim >= 0 && ndim <= kMaxDim, "tnn_ewise_compare: ndim %d > %d", ndim, kMaxDim);
    Dims d = collapse(ndim, shape, stride_a, stride_b);
    TNN_FLOAT_SWITCH(dtype, "tnn_ewise_compare", (compare_dispatch<T>(c, a, b, out_u8, d)));
}

int tnn_ewise_scalar(int op, const void* a, double s, int scalar_lhs, void* out, int64_t n,
                     int dtype) {
    TNN_NEED_INIT();
    if (scalar_lhs) {
        TNN_FLOAT_SWITCH(dtype, "tnn_ewise_scalar", (scalar_dispatch<T, true>(op, a, s, out, n)));
    }
    TNN_FLOAT_SWITCH(dtype, "tnn_ewise_scalar", (scalar_dispatch<T, false>(op, a, s, out, n)));
}

int tnn_compare_scalar(int c, const void* a, double s, void* out_u8, int64_t n, int dtype) {
    TNN_NEED_INIT();
#define CS(CODE) case CODE: TNN_FLOAT_SWITCH(dtype, "tnn_compare_scalar", \
        (launch_map<T, uint8_t>(a, out_u8, n, CmpScalarF<T, CODE>{(T)s})))
    switch (c) {
        CS(TNN_GT); CS(TNN_GE); CS(TNN_LT); CS(TNN_LE); CS(TNN_EQ); CS(TNN_NE);
    }
#undef CS
    tnn::set_error("tnn_compare_scalar: unknown comparison %d", c);
    return 2;
}

int tnn_ewise_unary(int op, const void* in, void* out, int64_t n, int dtype) {
    TNN_NEED_INIT();
    TNN_FLOAT_SWITCH(dtype, "tnn_ewise_unary", (unary_dispatch<T>(op, in, out, n)));
}

int tnn_clip(const void* in, int has_min, double vmin, int has_max, double vmax, void* out,
             int64_t n, int dtype) {
    TNN_NEED_INIT();
    TNN_FLOAT_SWITCH(dtype, "tnn_clip",
                     (launch_flat1<T>(in, out, n,
                                      ClipF<T>{(T)vmin, (T)vmax, has_min != 0, has_max != 0})));
}

int tnn_clip_bwd(const void* g, const void* x, int has_min, double vmin, int has_max, double vmax,
                 void* out, int64_t n, int dtype) {
    TNN_NEED_INIT();
    TNN_FLOAT_SWITCH(dtype, "tnn_clip_bwd",
                     (launch_flat2<T>(g, x, out, n,
                                      ClipBwdF<T>{(T)vmin, (T)vmax, has_min != 0, has_max != 0})));
}

int tnn_mul_mask(const void* g, const void* mask_u8, void* out, int64_t n, int dtype) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    switch (dtype) {
        case TNN_F32:
            hipLaunchKernelGGL((mul_mask_kernel<float>), tnn::stream_grid(n, kThreads), kThreads, 0,
                               tnn::stream(), (const float*)g, (const uint8_t*)mask_u8, (float*)out, n);
            break;
        case TNN_F64:
            hipLaunchKernelGGL((mul_mask_kernel<double>), tnn::stream_grid(n, kThreads), kThreads, 0,
                               tnn::stream(), (const double*)g, (const uint8_t*)mask_u8, (double*)out, n);
            break;
        default: tnn::set_error("tnn_mul_mask: dtype %d is not a float type", dtype); return 2;
    }
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_mul_signmask(const void* g, const void* y, void* out, int64_t n, int dtype) {
    TNN_NEED_INIT();
    TNN_FLOAT_SWITCH(dtype, "tnn_mul_signmask", (launch_flat2<T>(g, y, out, n, SignMaskF<T>{})));
}

int tnn_axpy(void* y, double alpha, const void* x, int64_t n, int dtype) {
    TNN_NEED_INIT();
    TNN_FLOAT_SWITCH(dtype, "tnn_axpy", (launch_flat2<T>(y, x, y, n, AxpyF<T>{(T)alpha})));
}

int tnn_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n) {
    TNN_NEED_INIT();
    switch (in_dtype) {
        case TNN_F32: return cast_from<float>(in, out, out_dtype, n);
        case TNN_F64: return cast_from<double>(in, out, out_dtype, n);
        case TNN_I64: return cast_from<int64_t>(in, out, out_dtype, n);
        case TNN_U8: return cast_from<uint8_t>(in, out, out_dtype, n);
    }
    tnn::set_error("tnn_cast: unknown input dtype %d", in_dtype);
    return 2;
}

int tnn_fill(void* dst, double value, int64_t n, int dtype) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    if (value == 0.0) return tnn_memset(dst, 0, (size_t)n * dtype_size(dtype));
    unsigned grid = tnn::stream_grid(n, kThreads);
    switch (dtype) {
        case TNN_F32: hipLaunchKernelGGL((fill_kernel<float>), grid, kThreads, 0, tnn::stream(), (float*)dst, n, (float)value); break;
        case TNN_F64: hipLaunchKernelGGL((fill_kernel<double>), grid, kThreads, 0, tnn::stream(), (double*)dst, n, value); break;
        case TNN_I64: hipLaunchKernelGGL((fill_kernel<int64_t>), grid, kThreads, 0, tnn::stream(), (int64_t*)dst, n, (int64_t)value); break;
        case TNN_U8: hipLaunchKernelGGL((fill_kernel<uint8_t>), grid, kThreads, 0, tnn::stream(), (uint8_t*)dst, n, (uint8_t)(value != 0.0)); break;
        default: tnn::set_error("tnn_fill: unknown dtype %d", dtype); return 2;
    }
    TNN_LAUNCH_OK();
    return 0;
}

#define TNN_ANY_SWITCH(dtype, fn_name, CALL)                                  \
    switch (dtype) {                                                          \
        case TNN_F32: { using T = float; return CALL; }                       \
        case TNN_F64: { using T = double; return CALL; }                      \
        case TNN_I64: { using T = int64_t; return CALL; }                     \
        case TNN_U8: { using T = uint8_t; return CALL; }                      \
        default: tnn::set_error(fn_name ": unknown dtype %d", dtype); return 2; \
    }

int tnn_strided_copy(const void* in, const int64_t* in_stride, void* out, int ndim,
                     const int64_t* shape, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(ndim >= 0 && ndim <= kMaxDim, "tnn_strided_copy: ndim %d > %d", ndim, kMaxDim);
    TNN_ANY_SWITCH(dtype, "tnn_strided_copy",
                   (strided_move<T>(in, out, in_stride, ndim, shape, false)));
}

int tnn_strided_scatter(const void* in, void* out, const int64_t* out_stride, int ndim,
                        const int64_t* shape, int dtype) {
    TNN_NEED_INIT();
    TNN_REQUIRE(ndim >= 0 && ndim <= kMaxDim, "tnn_strided_scatter: ndim %d > %d", ndim, kMaxDim);
    TNN_ANY_SWITCH(dtype, "tnn_strided_scatter",
                   (strided_move<T>(in, out, out_stride, ndim, shape, true)));
}

int tnn_gather_rows(const void* src, const void* idx_i64, void* out, int64_t n_idx,
                    int64_t row_elems, int64_t src_rows, int dtype) {
    TNN_NEED_INIT();
    TNN_ANY_SWITCH(dtype, "tnn_gather_rows",
                   (rows_move<T>(src, idx_i64, out, n_idx, row_elems, src_rows, false)));
}

int tnn_scatter_rows(const void* src, const void* idx_i64, void* dst, int64_t n_idx,
                     int64_t row_elems, int64_t dst_rows, int dtype) {
    TNN_NEED_INIT();
    TNN_ANY_SWITCH(dtype, "tnn_scatter_rows",
                   (rows_move<T>(src, idx_i64, dst, n_idx, row_elems, dst_rows, true)));
}

int tnn_gather_scalars(const void* ptrs_u64, void* out, int64_t n, int dtype) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    TNN_REQUIRE(ptrs_u64 && out, "tnn_gather_scalars: NULL argument");
    unsigned grid = tnn::stream_grid(n, kThreads);
    switch (dtype) {
        case TNN_F32: hipLaunchKernelGGL((gather_scalars_kernel<float>), grid, kThreads, 0, tnn::stream(), (const uint64_t*)ptrs_u64, (float*)out, n); break;
        case TNN_F64: hipLaunchKernelGGL((gather_scalars_kernel<double>), grid, kThreads, 0, tnn::stream(), (const uint64_t*)ptrs_u64, (double*)out, n); break;
        default: tnn::set_error("tnn_gather_scalars: dtype %d is not a float type", dtype); return 2;
    }
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_one_hot(const void* labels_i64, void* out, int64_t n, int64_t classes, int dtype) {
    TNN_NEED_INIT();
    if (n * classes <= 0) return 0;
    unsigned grid = tnn::stream_grid(n * classes, kThreads);
    switch (dtype) {
        case TNN_F32: hipLaunchKernelGGL((one_hot_kernel<float>), grid, kThreads, 0, tnn::stream(), (const int64_t*)labels_i64, (float*)out, n, classes); break;
        case TNN_F64: hipLaunchKernelGGL((one_hot_kernel<double>), grid, kThreads, 0, tnn::stream(), (const int64_t*)labels_i64, (double*)out, n, classes); break;
        default: tnn::set_error("tnn_one_hot: dtype %d is not a float type", dtype); return 2;
    }
    TNN_LAUNCH_OK();
    return 0;
}

}  // extern "C"
