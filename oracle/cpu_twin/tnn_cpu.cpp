// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
// CPU twin of include/tnn_hip.h: the same C-ABI implemented with plain scalar C++ loops on host
// memory ("device pointers" are malloc'ed host buffers).  It exists so that the host-side logic of
// tinynn-autograd_amd (DeviceArray / Tensor / ops / optimizers / data-parallel sharding) can be
// exercised by `pytest -m "not gpu"` in a container without a GPU.  Only tests/ may load it, through
// tinynn_autograd_amd._lib.install_test_twin(); the product loader never looks for it and fails
// loudly when libtnn_hip.so or a GPU is missing.  Each function restates the numpy expression of the
// reference it stands for (same citations as the header).  The whole-step trainer (tnn_mlp_*) is the
// product's own host code (csrc/tnn_mlp.cpp) compiled against these primitives.
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <functional>
#include <unordered_map>
#include <set>
#include <vector>

#include "tnn_hip.h"

namespace tnn {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace tnn

namespace {
bool g_ready = false;
std::unordered_map<void*, size_t> g_blocks;
int64_t g_live = 0, g_allocs = 0;

// "graph": a recorded list of closures replayed in order (captures the call sequence like a hipGraph)
struct Graph { std::vector<std::function<int()>> calls; };
Graph* g_capturing = nullptr;

#define REQ(cond, ...) do { if (!(cond)) { tnn::set_error(__VA_ARGS__); return 2; } } while (0)
#define NEED_INIT() REQ(g_ready, "tnn_init() has not been called")
// like hipStreamBeginCapture: while capturing, a primitive call is stored for replay and NOT executed
#define RECORD(...) do { if (g_capturing) { g_capturing->calls.push_back([=]() -> int { __VA_ARGS__; return 0; }); return 0; } } while (0)

constexpr int kMaxDim = 6;

template <typename F>
void nd_loop(int ndim, const int64_t* shape, F f) {
    int64_t n = 1;
    for (int k = 0; k < ndim; ++k) n *= shape[k];
    int64_t idx[kMaxDim] = {0};
    for (int64_t i = 0; i < n; ++i) {
        f(i, idx);
        for (int k = ndim - 1; k >= 0; --k) {
            if (++idx[k] < shape[k]) break;
            idx[k] = 0;
        }
    }
}
inline int64_t dot_idx(int ndim, const int64_t* idx, const int64_t* st) {
    int64_t o = 0;
    for (int k = 0; k < ndim; ++k) o += idx[k] * (st ? st[k] : 0);
    return o;
}

template <typename T> T bin(int op, T a, T b) {
    switch (op) {
        case TNN_ADD: return a + b;
        case TNN_SUB: return a - b;
        case TNN_MUL: return a * b;
        case TNN_DIV: return a / b;
        case TNN_POW: return (T)pow((double)a, (double)b);
        case TNN_MAX: return a >= b ? a : b;
        case TNN_MIN: return a <= b ? a : b;
    }
    return a;
}
template <typename T> uint8_t cmpf(int c, T a, T b) {
    switch (c) {
        case TNN_GT: return a > b;
        case TNN_GE: return a >= b;
        case TNN_LT: return a < b;
        case TNN_LE: return a <= b;
        case TNN_EQ: return a == b;
        case TNN_NE: return a != b;
    }
    return 0;
}
template <typename T> T una(int op, T a) {
    switch (op) {
        case TNN_NEG: return -a;
        case TNN_EXP: return (T)exp((double)a);
        case TNN_LOG: return (T)log((double)a);
        case TNN_SQRT: return (T)sqrt((double)a);
        case TNN_SQUARE: return a * a;
        case TNN_ABS: return a < 0 ? -a : a;
        case TNN_RECIP: return T(1) / a;
        case TNN_SIGMOID: return (T)(1.0 / (1.0 + exp(-(double)a)));
        case TNN_TANH: return (T)tanh((double)a);
    }
    return a;
}

size_t dsize(int dtype) { return dtype == TNN_F32 ? 4 : dtype == TNN_U8 ? 1 : dtype == TNN_BF16 ? 2 : 8; }

#define FLOAT_SWITCH(dtype, fn, ...)                                          \
    switch (dtype) {                                                          \
        case TNN_F32: { using T = float; __VA_ARGS__; break; }                       \
        case TNN_F64: { using T = double; __VA_ARGS__; break; }                      \
        default: tnn::set_error(fn ": dtype %d is not a float type", dtype); return 2; \
    }
#define ANY_SWITCH(dtype, fn, ...)                                            \
    switch (dtype) {                                                          \
        case TNN_F32: { using T = float; __VA_ARGS__; break; }                       \
        case TNN_F64: { using T = double; __VA_ARGS__; break; }                      \
        case TNN_I64: { using T = int64_t; __VA_ARGS__; break; }                     \
        case TNN_U8: { using T = uint8_t; __VA_ARGS__; break; }                      \
        default: tnn::set_error(fn ": unknown dtype %d", dtype); return 2;    \
    }

template <typename T>
void gemm_ref(int tA, int tB, int64_t M, int64_t N, int64_t K, const T* A, int64_t lda, const T* B,
              int64_t ldb, std::vector<T>& acc) {
    acc.assign((size_t)(M * N), T(0));
    for (int64_t i = 0; i < M; ++i)
        for (int64_t k = 0; k < K; ++k) {
            T a = tA ? A[k * lda + i] : A[i * lda + k];
            for (int64_t j = 0; j < N; ++j) {
                T b = tB ? B[j * ldb + k] : B[k * ldb + j];
                acc[(size_t)(i * N + j)] += a * b;
            }
        }
}

struct Strides { int64_t v[kMaxDim]; bool null; };
Strides keep(const int64_t* s, int nd) {
    Strides r;
    r.null = s == nullptr;
    for (int k = 0; k < kMaxDim; ++k) r.v[k] = (s && k < nd) ? s[k] : 0;
    return r;
}
}  // namespace

extern "C" {

const char* tnn_last_error(void) { return tnn::g_err; }
int tnn_backend_kind(void) { return 2; }
int tnn_init(int) { g_ready = true; return 0; }
int tnn_shutdown(void) { return 0; }

int tnn_device_props(int* cu, int* clk, int64_t* hbm, char* name, int n) {
    if (cu) *cu = 0;
    if (clk) *clk = 0;
    if (hbm) *hbm = 0;
    if (name && n > 0) snprintf(name, n, "cpu-twin (tests only)");
    return 0;
}
static std::set<void*> g_graph_blocks;
int tnn_malloc(size_t bytes, void** out) {
    NEED_INIT();
    void* p = nullptr;
    if (posix_memalign(&p, 64, bytes ? bytes : 1)) { tnn::set_error("tnn_malloc: out of memory"); return 1; }
    g_blocks[p] = bytes;
    if (g_capturing) g_graph_blocks.insert(p);      // handed out during a capture: owned by that graph, like in the HIP library
    g_live += (int64_t)bytes;
    g_allocs++;
    *out = p;
    return 0;
}
int tnn_free(void* p) {
    if (!p) return 0;
    auto it = g_blocks.find(p);
    REQ(it != g_blocks.end(), "tnn_free: %p was not allocated by tnn_malloc", p);
    g_live -= (int64_t)it->second;
    g_blocks.erase(it);
    // buffers touched by a recorded graph stay valid (leaked in tests): freed while a capture is open, or allocated during one
    // (the HIP library never recycles a graph-owned buffer outside its graph either)
    if (!g_capturing && !g_graph_blocks.count(p)) free(p);
    return 0;
}
int tnn_pool_stats(int64_t* live, int64_t* cached, int64_t* allocs) {
    if (live) *live = g_live;
    if (cached) *cached = 0;
    if (allocs) *allocs = g_allocs;
    return 0;
}
int tnn_pool_trim(void) { return 0; }
int tnn_memcpy_h2d(void* d, const void* s, size_t n) { NEED_INIT(); memcpy(d, s, n); return 0; }
int tnn_memcpy_d2h(void* d, const void* s, size_t n) { NEED_INIT(); memcpy(d, s, n); return 0; }
int tnn_memcpy_d2d(void* d, const void* s, size_t n) { NEED_INIT(); RECORD(memmove(d, s, n)); memmove(d, s, n); return 0; }
int tnn_memset(void* d, int b, size_t n) { NEED_INIT(); RECORD(memset(d, b, n)); memset(d, b, n); return 0; }
int tnn_stream_sync(void) { return 0; }
int tnn_event_create(void** ev) { *ev = malloc(sizeof(double)); return 0; }
int tnn_event_record(void* ev) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    *(double*)ev = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    return 0;
}
int tnn_event_elapsed_ms(void* a, void* b, float* ms) { *ms = (float)(*(double*)b - *(double*)a); return 0; }
int tnn_event_destroy(void* ev) { free(ev); return 0; }

int tnn_graph_capture_begin(void) {
    REQ(!g_capturing, "tnn_graph_capture_begin: a capture is already open");
    g_capturing = new Graph();
    return 0;
}
int tnn_graph_capture_end(void** ge) {
    REQ(g_capturing, "tnn_graph_capture_end: no capture is open");
    *ge = g_capturing;
    g_capturing = nullptr;
    return 0;
}
int tnn_graph_launch(void* ge) {
    REQ(ge, "tnn_graph_launch: invalid graph");
    for (auto& c : ((Graph*)ge)->calls)
        if (int rc = c()) return rc;
    return 0;
}
int tnn_graph_destroy(void* ge) { delete (Graph*)ge; return 0; }

int tnn_fill(void* dst, double value, int64_t n, int dtype) {
    NEED_INIT();
    RECORD(tnn_fill(dst, value, n, dtype));
    ANY_SWITCH(dtype, "tnn_fill", { T* o = (T*)dst; for (int64_t i = 0; i < n; ++i) o[i] = (T)value; });
    return 0;
}

// ---- GEMM ----
static int gemm_common(int tA, int tB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                       const void* B, int64_t ldb, void* C, int64_t ldc, int dtype, int epi,
                       double alpha, double beta, const void* bias, int act, int relu_sign,
                       const void* Y, int64_t ldy) {
    FLOAT_SWITCH(dtype, "tnn_gemm", {
        std::vector<T> acc;
        gemm_ref<T>(tA, tB, M, N, K, (const T*)A, lda, (const T*)B, ldb, acc);
        T* c = (T*)C;
        for (int64_t i = 0; i < M; ++i)
            for (int64_t j = 0; j < N; ++j) {
                T v = acc[(size_t)(i * N + j)];
                if (epi == 0) {
                    v = (T)alpha * v;
                    if (beta != 0.0) v += (T)beta * c[i * ldc + j];
                } else if (epi == 1) {
                    v += bias ? ((const T*)bias)[j] : T(0);
                    if (act == TNN_ACT_RELU) {
                        if (relu_sign) v = v < 0 ? (T)-0.0 : (T)fabs((double)v);
                        else v = v < 0 ? T(0) : v;
                    }
                } else {
                    v = signbit((double)((const T*)Y)[i * ldy + j]) ? T(0) : v;
                }
                c[i * ldc + j] = v;
            }
    });
    return 0;
}
int tnn_gemm(int tA, int tB, int64_t M, int64_t N, int64_t K, double alpha, const void* A, int64_t lda,
             const void* B, int64_t ldb, double beta, void* C, int64_t ldc, int dtype) {
    NEED_INIT();
    REQ(lda >= (tA ? M : K) && ldb >= (tB ? K : N) && ldc >= N, "tnn_gemm: leading dimension too small");
    RECORD(tnn_gemm(tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, dtype));
    return gemm_common(tA, tB, M, N, K, A, lda, B, ldb, C, ldc, dtype, 0, alpha, beta, nullptr, 0, 0, nullptr, 0);
}
int tnn_gemm_bias_act(int tA, int tB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                      const void* B, int64_t ldb, const void* bias, int act, int relu_sign, void* C,
                      int64_t ldc, int dtype) {
    NEED_INIT();
    RECORD(tnn_gemm_bias_act(tA, tB, M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, dtype));
    return gemm_common(tA, tB, M, N, K, A, lda, B, ldb, C, ldc, dtype, 1, 1, 0, bias, act, relu_sign, nullptr, 0);
}
int tnn_gemm_mask(int tA, int tB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                  const void* B, int64_t ldb, const void* Y, int64_t ldy, void* C, int64_t ldc, int dtype) {
    NEED_INIT();
    RECORD(tnn_gemm_mask(tA, tB, M, N, K, A, lda, B, ldb, Y, ldy, C, ldc, dtype));
    return gemm_common(tA, tB, M, N, K, A, lda, B, ldb, C, ldc, dtype, 2, 1, 0, nullptr, 0, 0, Y, ldy);
}

int tnn_gemm_tn_colsum(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* G, int64_t ldg,
                       void* dW, int64_t ldc, void* db, int dtype) {
    NEED_INIT();
    REQ(ldg == N || db == nullptr, "tnn_gemm_tn_colsum: the column sum needs a dense G (ldg == N)");
    RECORD(tnn_gemm_tn_colsum(M, N, K, A, lda, G, ldg, dW, ldc, db, dtype));
    if (int rc = gemm_common(1, 0, M, N, K, A, lda, G, ldg, dW, ldc, dtype, 0, 1.0, 0.0, nullptr, 0, 0, nullptr, 0)) return rc;
    return db ? tnn_reduce(TNN_RSUM, G, db, 1, K, N, dtype) : 0;
}

int tnn_adam_ex(void*, const void*, void*, void*, int64_t, double, double, double, double, void*, void*, int, int, const void*,
                void*);
int tnn_gemm_tn_adam(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* G, int64_t ldg, void* g_out,
                     void* p, void* m, void* v, double lr, double b1, double b2, double eps, const void* pows, int dtype) {
    NEED_INIT();
    REQ(p && m && v && pows, "tnn_gemm_tn_adam: p, m, v and pows are required");
    RECORD(tnn_gemm_tn_adam(M, N, K, A, lda, G, ldg, g_out, p, m, v, lr, b1, b2, eps, pows, dtype));
    std::vector<char> scratch;
    void* gw = g_out;
    if (!gw) { scratch.resize((size_t)(M * N) * (dtype == TNN_F64 ? 8 : 4)); gw = scratch.data(); }
    if (int rc = tnn_gemm_tn_colsum(M, N, K, A, lda, G, ldg, gw, N, nullptr, dtype)) return rc;
    return tnn_adam_ex(p, gw, m, v, M * N, lr, b1, b2, eps, const_cast<void*>(pows), nullptr, dtype, 0, nullptr, nullptr);
}
int tnn_gemm_tn_adam_bias(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* G, int64_t ldg, void* g_out,
                          void* p, void* m, void* v, void* db, void* pb, void* mb, void* vb, double lr, double b1, double b2,
                          double eps, const void* pows, int dtype) {
    NEED_INIT();
    REQ((pb == nullptr) == (mb == nullptr) && (pb == nullptr) == (vb == nullptr) && (pb == nullptr || db != nullptr),
        "tnn_gemm_tn_adam_bias: pb / mb / vb go together and need db");
    RECORD(tnn_gemm_tn_adam_bias(M, N, K, A, lda, G, ldg, g_out, p, m, v, db, pb, mb, vb, lr, b1, b2, eps, pows, dtype));
    if (int rc = tnn_gemm_tn_adam(M, N, K, A, lda, G, ldg, g_out, p, m, v, lr, b1, b2, eps, pows, dtype)) return rc;
    if (!db) return 0;
    if (int rc = tnn_reduce(TNN_RSUM, G, db, 1, K, N, dtype)) return rc;
    if (!pb) return 0;
    return tnn_adam_ex(pb, db, mb, vb, N, lr, b1, b2, eps, const_cast<void*>(pows), nullptr, dtype, 0, nullptr, nullptr);
}
int tnn_dense_bwd(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz, const void* w,
                  void* dw, void* db, void* dx, const void* mask_src, int dtype) {
    NEED_INIT();
    REQ(dx == nullptr || mask_src != nullptr, "tnn_dense_bwd: dx needs mask_src");
    if (int rc = tnn_gemm_tn_colsum(n_in, n_out, rows, x, n_in, dz, n_out, dw, n_out, db, dtype)) return rc;
    if (dx) return tnn_gemm_mask(0, 1, rows, n_in, n_out, dz, n_out, w, n_out, mask_src, n_in, dx, n_in, dtype);
    return 0;
}

// ---- elementwise ----
int tnn_ewise_binary(int op, const void* a, const int64_t* sa, const void* b, const int64_t* sb,
                     void* out, int ndim, const int64_t* shape, int dtype) {
    NEED_INIT();
    REQ(ndim >= 0 && ndim <= kMaxDim, "tnn_ewise_binary: ndim %d", ndim);
    REQ(op >= TNN_ADD && op <= TNN_MIN, "tnn_ewise_binary: unknown op %d", op);
    Strides ka = keep(sa, ndim), kb = keep(sb, ndim), ks = keep(shape, ndim);
    RECORD(tnn_ewise_binary(op, a, ka.null ? nullptr : ka.v, b, kb.null ? nullptr : kb.v, out, ndim, ks.v, dtype));
    FLOAT_SWITCH(dtype, "tnn_ewise_binary", {
        nd_loop(ndim, shape, [&](int64_t i, const int64_t* idx) {
            ((T*)out)[i] = bin<T>(op, ((const T*)a)[dot_idx(ndim, idx, sa)], ((const T*)b)[dot_idx(ndim, idx, sb)]);
        });
    });
    return 0;
}
int tnn_ewise_compare(int c, const void* a, const int64_t* sa, const void* b, const int64_t* sb,
                      void* out, int ndim, const int64_t* shape, int dtype) {
    NEED_INIT();
    REQ(ndim >= 0 && ndim <= kMaxDim, "tnn_ewise_compare: ndim %d", ndim);
    REQ(c >= TNN_GT && c <= TNN_NE, "tnn_ewise_compare: unknown comparison %d", c);
    Strides ka = keep(sa, ndim), kb = keep(sb, ndim), ks = keep(shape, ndim);
    RECORD(tnn_ewise_compare(c, a, ka.null ? nullptr : ka.v, b, kb.null ? nullptr : kb.v, out, ndim, ks.v, dtype));
    FLOAT_SWITCH(dtype, "tnn_ewise_compare", {
        nd_loop(ndim, shape, [&](int64_t i, const int64_t* idx) {
            ((uint8_t*)out)[i] = cmpf<T>(c, ((const T*)a)[dot_idx(ndim, idx, sa)], ((const T*)b)[dot_idx(ndim, idx, sb)]);
        });
    });
    return 0;
}
int tnn_ewise_scalar(int op, const void* a, double s, int lhs, void* out, int64_t n, int dtype) {
    NEED_INIT();
    REQ(op >= TNN_ADD && op <= TNN_MIN, "tnn_ewise_scalar: unknown op %d", op);
    RECORD(tnn_ewise_scalar(op, a, s, lhs, out, n, dtype));
    FLOAT_SWITCH(dtype, "tnn_ewise_scalar", {
        for (int64_t i = 0; i < n; ++i) {
            T x = ((const T*)a)[i];
            T r;
            if (op == TNN_POW && !lhs && s == 2.0) r = x * x;
            else if (op == TNN_POW && !lhs && s == 0.5) r = (T)sqrt((double)x);
            else r = lhs ? bin<T>(op, (T)s, x) : bin<T>(op, x, (T)s);
            ((T*)out)[i] = r;
        }
    });
    return 0;
}
int tnn_compare_scalar(int c, const void* a, double s, void* out, int64_t n, int dtype) {
    NEED_INIT();
    REQ(c >= TNN_GT && c <= TNN_NE, "tnn_compare_scalar: unknown comparison %d", c);
    RECORD(tnn_compare_scalar(c, a, s, out, n, dtype));
    FLOAT_SWITCH(dtype, "tnn_compare_scalar", {
        for (int64_t i = 0; i < n; ++i) ((uint8_t*)out)[i] = cmpf<T>(c, ((const T*)a)[i], (T)s);
    });
    return 0;
}
int tnn_ewise_unary(int op, const void* in, void* out, int64_t n, int dtype) {
    NEED_INIT();
    REQ(op >= TNN_NEG && op <= TNN_COPY, "tnn_ewise_unary: unknown op %d", op);
    RECORD(tnn_ewise_unary(op, in, out, n, dtype));
    FLOAT_SWITCH(dtype, "tnn_ewise_unary", {
        for (int64_t i = 0; i < n; ++i) ((T*)out)[i] = una<T>(op, ((const T*)in)[i]);
    });
    return 0;
}
int tnn_clip(const void* in, int hmin, double vmin, int hmax, double vmax, void* out, int64_t n, int dtype) {
    NEED_INIT();
    RECORD(tnn_clip(in, hmin, vmin, hmax, vmax, out, n, dtype));
    FLOAT_SWITCH(dtype, "tnn_clip", {
        for (int64_t i = 0; i < n; ++i) {
            T x = ((const T*)in)[i];
            if (hmin && x < (T)vmin) x = (T)vmin;
            if (hmax && x > (T)vmax) x = (T)vmax;
            ((T*)out)[i] = x;
        }
    });
    return 0;
}
int tnn_clip_bwd(const void* g, const void* x, int hmin, double vmin, int hmax, double vmax, void* out,
                 int64_t n, int dtype) {
    NEED_INIT();
    RECORD(tnn_clip_bwd(g, x, hmin, vmin, hmax, vmax, out, n, dtype));
    FLOAT_SWITCH(dtype, "tnn_clip_bwd", {
        for (int64_t i = 0; i < n; ++i) {
            T xv = ((const T*)x)[i];
            bool keepv = (!hmin || xv >= (T)vmin) && (!hmax || xv <= (T)vmax);
            ((T*)out)[i] = keepv ? ((const T*)g)[i] : T(0);
        }
    });
    return 0;
}
int tnn_mul_signmask(const void* g, const void* y, void* out, int64_t n, int dtype) {
    NEED_INIT();
    RECORD(tnn_mul_signmask(g, y, out, n, dtype));
    FLOAT_SWITCH(dtype, "tnn_mul_signmask", {
        for (int64_t i = 0; i < n; ++i) ((T*)out)[i] = std::signbit(((const T*)y)[i]) ? T(0) : ((const T*)g)[i];
    });
    return 0;
}
int tnn_mul_mask(const void* g, const void* mask, void* out, int64_t n, int dtype) {
    NEED_INIT();
    RECORD(tnn_mul_mask(g, mask, out, n, dtype));
    FLOAT_SWITCH(dtype, "tnn_mul_mask", {
        for (int64_t i = 0; i < n; ++i) ((T*)out)[i] = ((const uint8_t*)mask)[i] ? ((const T*)g)[i] : T(0);
    });
    return 0;
}
int tnn_axpy(void* y, double alpha, const void* x, int64_t n, int dtype) {
    NEED_INIT();
    RECORD(tnn_axpy(y, alpha, x, n, dtype));
    FLOAT_SWITCH(dtype, "tnn_axpy", {
        for (int64_t i = 0; i < n; ++i) ((T*)y)[i] = ((T*)y)[i] + (T)alpha * ((const T*)x)[i];
    });
    return 0;
}
int tnn_cast(const void* in, int idt, void* out, int odt, int64_t n) {
    NEED_INIT();
    RECORD(tnn_cast(in, idt, out, odt, n));
    for (int64_t i = 0; i < n; ++i) {
        double v;
        switch (idt) {
            case TNN_F32: v = ((const float*)in)[i]; break;
            case TNN_F64: v = ((const double*)in)[i]; break;
            case TNN_I64: v = (double)((const int64_t*)in)[i]; break;
            case TNN_U8: v = ((const uint8_t*)in)[i]; break;
            default: tnn::set_error("tnn_cast: unknown input dtype %d", idt); return 2;
        }
        switch (odt) {
            case TNN_F32: ((float*)out)[i] = (float)v; break;
            case TNN_F64: ((double*)out)[i] = v; break;
            case TNN_I64: ((int64_t*)out)[i] = idt == TNN_I64 ? ((const int64_t*)in)[i] : (int64_t)v; break;
            case TNN_U8: ((uint8_t*)out)[i] = v != 0.0; break;
            default: tnn::set_error("tnn_cast: unknown output dtype %d", odt); return 2;
        }
    }
    return 0;
}

// ---- reductions ----
int tnn_reduce(int rop, const void* in, void* out, int64_t outer, int64_t red, int64_t inner, int dtype) {
    NEED_INIT();
    REQ(rop >= TNN_RSUM && rop <= TNN_RMIN, "tnn_reduce: unknown reduction %d", rop);
    REQ(red > 0 || rop == TNN_RSUM, "tnn_reduce: max/min of an empty axis");
    RECORD(tnn_reduce(rop, in, out, outer, red, inner, dtype));
    FLOAT_SWITCH(dtype, "tnn_reduce", {
        for (int64_t o = 0; o < outer; ++o)
            for (int64_t i = 0; i < inner; ++i) {
                double acc = rop == TNN_RSUM ? 0.0 : rop == TNN_RMAX ? -INFINITY : INFINITY;
                for (int64_t r = 0; r < red; ++r) {
                    double v = ((const T*)in)[(o * red + r) * inner + i];
                    acc = rop == TNN_RSUM ? acc + v : rop == TNN_RMAX ? (v > acc ? v : acc) : (v < acc ? v : acc);
                }
                ((T*)out)[o * inner + i] = (T)acc;
            }
    });
    return 0;
}
int tnn_argmax_rows(const void* in, void* out, int64_t rows, int64_t cols, int dtype) {
    NEED_INIT();
    REQ(cols > 0 || rows == 0, "tnn_argmax_rows: empty rows");
    RECORD(tnn_argmax_rows(in, out, rows, cols, dtype));
    FLOAT_SWITCH(dtype, "tnn_argmax_rows", {
        for (int64_t r = 0; r < rows; ++r) {
            const T* p = (const T*)in + r * cols;
            int64_t bi = 0;
            for (int64_t k = 1; k < cols; ++k) if (p[k] > p[bi]) bi = k;
            ((int64_t*)out)[r] = bi;
        }
    });
    return 0;
}

// ---- data movement ----
int tnn_strided_copy(const void* in, const int64_t* st, void* out, int ndim, const int64_t* shape, int dtype) {
    NEED_INIT();
    REQ(ndim >= 0 && ndim <= kMaxDim, "tnn_strided_copy: ndim %d", ndim);
    Strides k1 = keep(st, ndim), ks = keep(shape, ndim);
    RECORD(tnn_strided_copy(in, k1.v, out, ndim, ks.v, dtype));
    ANY_SWITCH(dtype, "tnn_strided_copy", {
        nd_loop(ndim, shape, [&](int64_t i, const int64_t* idx) { ((T*)out)[i] = ((const T*)in)[dot_idx(ndim, idx, st)]; });
    });
    return 0;
}
int tnn_strided_scatter(const void* in, void* out, const int64_t* st, int ndim, const int64_t* shape, int dtype) {
    NEED_INIT();
    REQ(ndim >= 0 && ndim <= kMaxDim, "tnn_strided_scatter: ndim %d", ndim);
    Strides k1 = keep(st, ndim), ks = keep(shape, ndim);
    RECORD(tnn_strided_scatter(in, out, k1.v, ndim, ks.v, dtype));
    ANY_SWITCH(dtype, "tnn_strided_scatter", {
        nd_loop(ndim, shape, [&](int64_t i, const int64_t* idx) { ((T*)out)[dot_idx(ndim, idx, st)] = ((const T*)in)[i]; });
    });
    return 0;
}
int tnn_gather_rows(const void* src, const void* idx, void* out, int64_t n, int64_t re, int64_t rows, int dtype) {
    NEED_INIT();
    RECORD(tnn_gather_rows(src, idx, out, n, re, rows, dtype));
    size_t es = dsize(dtype);
    for (int64_t i = 0; i < n; ++i) {
        int64_t j = ((const int64_t*)idx)[i];
        if (j < 0) j += rows;
        if (j < 0 || j >= rows) continue;
        memcpy((char*)out + i * re * es, (const char*)src + j * re * es, re * es);
    }
    return 0;
}
int tnn_scatter_rows(const void* src, const void* idx, void* dst, int64_t n, int64_t re, int64_t rows, int dtype) {
    NEED_INIT();
    RECORD(tnn_scatter_rows(src, idx, dst, n, re, rows, dtype));
    size_t es = dsize(dtype);
    for (int64_t i = 0; i < n; ++i) {
        int64_t j = ((const int64_t*)idx)[i];
        if (j < 0) j += rows;
        if (j < 0 || j >= rows) continue;
        memcpy((char*)dst + j * re * es, (const char*)src + i * re * es, re * es);
    }
    return 0;
}
int tnn_gather_scalars(const void* ptrs, void* out, int64_t n, int dtype) {
    NEED_INIT();
    REQ(dtype == TNN_F32 || dtype == TNN_F64, "tnn_gather_scalars: dtype %d is not a float type", dtype);
    RECORD(tnn_gather_scalars(ptrs, out, n, dtype));
    for (int64_t i = 0; i < n; ++i) {
        const void* src = reinterpret_cast<const void*>(((const uint64_t*)ptrs)[i]);
        if (dtype == TNN_F32) ((float*)out)[i] = *(const float*)src;
        else ((double*)out)[i] = *(const double*)src;
    }
    return 0;
}
int tnn_one_hot(const void* labels, void* out, int64_t n, int64_t classes, int dtype) {
    NEED_INIT();
    RECORD(tnn_one_hot(labels, out, n, classes, dtype));
    FLOAT_SWITCH(dtype, "tnn_one_hot", {
        for (int64_t i = 0; i < n; ++i)
            for (int64_t c = 0; c < classes; ++c) ((T*)out)[i * classes + c] = ((const int64_t*)labels)[i] == c ? T(1) : T(0);
    });
    return 0;
}

// ---- fused ----
int tnn_bias_act(const void* x, const void* bias, int act, void* y, int64_t M, int64_t N, int dtype) {
    NEED_INIT();
    RECORD(tnn_bias_act(x, bias, act, y, M, N, dtype));
    FLOAT_SWITCH(dtype, "tnn_bias_act", {
        for (int64_t i = 0; i < M; ++i)
            for (int64_t j = 0; j < N; ++j) {
                T v = ((const T*)x)[i * N + j] + ((const T*)bias)[j];
                if (act == TNN_ACT_RELU && v < 0) v = 0;
                ((T*)y)[i * N + j] = v;
            }
    });
    return 0;
}
int tnn_softmax_nll_stats(const void* z, int64_t m, int64_t c, void* stats, int dtype) {
    NEED_INIT();
    REQ(m * c > 0, "tnn_softmax_nll_stats: empty logits");
    RECORD(tnn_softmax_nll_stats(z, m, c, stats, dtype));
    FLOAT_SWITCH(dtype, "tnn_softmax_nll_stats", {
        double mx = -INFINITY;
        for (int64_t i = 0; i < m * c; ++i) mx = fmax(mx, (double)((const T*)z)[i]);
        double s = 0;
        for (int64_t i = 0; i < m * c; ++i) s += exp((double)((const T*)z)[i] - mx);
        ((T*)stats)[0] = (T)mx;
        ((T*)stats)[1] = (T)s;
    });
    return 0;
}
int tnn_lse_merge(const void* all, int n, void* stats, int dtype) {
    NEED_INIT();
    REQ(n > 0, "tnn_lse_merge: n_shards %d", n);
    RECORD(tnn_lse_merge(all, n, stats, dtype));
    FLOAT_SWITCH(dtype, "tnn_lse_merge", {
        double mx = -INFINITY;
        for (int i = 0; i < n; ++i) mx = fmax(mx, (double)((const T*)all)[2 * i]);
        double s = 0;
        for (int i = 0; i < n; ++i) s += (double)((const T*)all)[2 * i + 1] * exp((double)((const T*)all)[2 * i] - mx);
        ((T*)stats)[0] = (T)mx;
        ((T*)stats)[1] = (T)s;
    });
    return 0;
}
int tnn_softmax_nll_fwd_bwd(const void* z, const void* y, int64_t m, int64_t c, int64_t mg, const void* stats,
                            void* loss_out, void* dz, int dtype) {
    NEED_INIT();
    REQ(m > 0 && c > 0 && mg > 0, "tnn_softmax_nll_fwd_bwd: empty batch");
    RECORD(tnn_softmax_nll_fwd_bwd(z, y, m, c, mg, stats, loss_out, dz, dtype));
    FLOAT_SWITCH(dtype, "tnn_softmax_nll_fwd_bwd", {
        double M = ((const T*)stats)[0], S = ((const T*)stats)[1], loss = 0;
        for (int64_t r = 0; r < m; ++r) {
            double q = 0;
            for (int64_t k = 0; k < c; ++k) q += exp((double)((const T*)z)[r * c + k] - M) * (double)((const T*)y)[r * c + k];
            loss += (log(S) - log(q)) / (double)mg;
            if (dz)
                for (int64_t k = 0; k < c; ++k) {
                    double e = exp((double)((const T*)z)[r * c + k] - M);
                    ((T*)dz)[r * c + k] = (T)(e / S - e * (double)((const T*)y)[r * c + k] / q / (double)mg);
                }
        }
        if (loss_out) ((T*)loss_out)[0] = (T)loss;
    });
    return 0;
}
int tnn_softmax_nll_fused(const void* z, const void* y, int64_t m, int64_t c, void* stats_out, void* loss_out,
                          void* dz, int dtype) {
    NEED_INIT();
    REQ(m > 0 && c > 0, "tnn_softmax_nll_fused: empty batch");
    RECORD(tnn_softmax_nll_fused(z, y, m, c, stats_out, loss_out, dz, dtype));
    double tmp[2];
    void* st = stats_out ? stats_out : (void*)tmp;
    if (int rc = tnn_softmax_nll_stats(z, m, c, st, dtype)) return rc;
    return tnn_softmax_nll_fwd_bwd(z, y, m, c, m, st, loss_out, dz, dtype);
}
int tnn_mlp_head(int64_t rows, int64_t nh, int64_t nc, const void* a, const void* w, const void* b, const void* y,
                 void* logits, void* dz, void* stats, void* loss, void* dw, void* db, void* da, int dtype) {
    NEED_INIT();
    REQ(logits && dz && dw && db, "tnn_mlp_head: logits, dz, dw and db buffers are required");
    if (int rc = tnn_gemm_bias_act(0, 0, rows, nc, nh, a, nh, w, nc, b, TNN_ACT_NONE, 0, logits, nc, dtype)) return rc;
    if (int rc = tnn_softmax_nll_fused(logits, y, rows, nc, stats, loss, dz, dtype)) return rc;
    return tnn_dense_bwd(rows, nh, nc, a, dz, w, dw, db, da, a, dtype);
}
int tnn_mlp_head_fits(int64_t rows, int64_t nh, int64_t nc, int dtype, int* fits) {
    REQ(fits != nullptr, "tnn_mlp_head_fits: fits is NULL");
    *fits = (dtype == TNN_F32 && nc == 10 && nh == 128 && rows >= 1 && rows <= 128) ? 1 : 0;    // the HIP kernel's shapes
    return 0;
}
int tnn_mlp_head_bwd_reserve(int64_t, int64_t, int64_t, int64_t) { return 0; }
int tnn_mlp_head_bwd_fits(int64_t rows, int64_t n_in, int64_t nh, int64_t nc, int dtype, int* fits) {
    REQ(fits != nullptr, "tnn_mlp_head_bwd_fits: fits is NULL");
    *fits = (dtype == TNN_F32 && rows >= 1 && rows <= 128 && nh % 16 == 0 && nh >= 16 && nh <= 256 && nc >= 1 && nc <= 16 &&
             n_in >= 16 && n_in % 16 == 0) ? 1 : 0;                                   // the HIP kernels' shapes
    return 0;
}
int tnn_softmax_nll_fused_tick(const void*, const void*, int64_t, int64_t, int64_t, int, void*, void*, void*, int, void*,
                               double, double);
int tnn_mlp_head_tick(int64_t rows, int64_t nh, int64_t nc, const void* a, const void* w, const void* b, const void* y,
                      const void* zpart, void* logits, void* dz, void* stats, void* loss, void* dw, void* db, void* da,
                      int dtype, void* pows, double b1, double b2) {
    NEED_INIT();
    REQ(logits && dz && dw && db, "cpu twin: tnn_mlp_head_tick needs logits and dz scratch");
    REQ(dtype == TNN_F32, "cpu twin: tnn_mlp_head_tick is f32 only");
    RECORD(tnn_mlp_head_tick(rows, nh, nc, a, w, b, y, zpart, logits, dz, stats, loss, dw, db, da, dtype, pows, b1, b2));
    if (zpart) {      // logits = bias + the tiles' partial sums (what the HIP kernel does with them)
        const int64_t tiles = (nh + 15) / 16;
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t c = 0; c < nc; ++c) {
                float s = 0.f;
                for (int64_t tn = 0; tn < tiles; ++tn) s += ((const float*)zpart)[(tn * rows + r) * nc + c];
                ((float*)logits)[r * nc + c] = s + ((const float*)b)[c];
            }
    } else if (int rc = tnn_gemm_bias_act(0, 0, rows, nc, nh, a, nh, w, nc, b, TNN_ACT_NONE, 0, logits, nc, dtype)) {
        return rc;
    }
    if (int rc = tnn_softmax_nll_fused_tick(logits, y, rows, nc, rows, 0, stats, loss, dz, dtype, pows, b1, b2)) return rc;
    return tnn_dense_bwd(rows, nh, nc, a, dz, w, dw, db, da, a, dtype);
}
int tnn_mlp_head_bwd_tick(int64_t rows, int64_t n_in, int64_t nh, int64_t nc, const void* x, const void* w1, const void* a,
                          const void* w, const void* b, const void* y, const void* zpart, void* logits, void* dz, void* stats,
                          void* loss, void* dw, void* db, void* dw1, void* db1, void* dx, int dtype, void* pows, double b1,
                          double b2) {
    NEED_INIT();
    REQ(x && w1 && zpart && logits && dz && dw && db && dw1 && db1 && dx, "cpu twin: tnn_mlp_head_bwd_tick needs every buffer");
    REQ(dtype == TNN_F32 && n_in % 16 == 0, "cpu twin: tnn_mlp_head_bwd_tick is f32 only, n_in % 16 == 0");
    RECORD(tnn_mlp_head_bwd_tick(rows, n_in, nh, nc, x, w1, a, w, b, y, zpart, logits, dz, stats, loss, dw, db, dw1, db1, dx,
                                 dtype, pows, b1, b2));
    std::vector<float> da((size_t)(rows * nh));              // the hidden layer's dz: never materialised by the HIP kernel
    if (int rc = tnn_mlp_head_tick(rows, nh, nc, a, w, b, y, zpart, logits, dz, stats, loss, dw, db, da.data(), dtype, pows, b1, b2))
        return rc;
    return tnn_dense_bwd(rows, n_in, nh, x, da.data(), w1, dw1, db1, dx, x, dtype);
}
static void twin_logits_from_partials(int64_t rows, int64_t nh, int64_t nc, const void* b, const void* zpart, float* logits) {
    const int64_t tiles = (nh + 15) / 16;
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < nc; ++c) {
            float s = 0.f;
            for (int64_t tn = 0; tn < tiles; ++tn) s += ((const float*)zpart)[(tn * rows + r) * nc + c];
            logits[r * nc + c] = s + ((const float*)b)[c];
        }
}
int tnn_dense_fwd_head_partials(int64_t, int64_t, int64_t, const void*, int64_t, const void*, int64_t, const void*, int, int, void*,
                                int64_t, const void*, int64_t, void*, int);
int tnn_dense_fwd_head_partials_stats(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                      const void* bias, int act, int relu_sign, void* C, int64_t ldc, const void* head_w,
                                      int64_t head_c, void* head_z, const void* head_b, const void* y, void* ticket, void* out_pair,
                                      int exchange, int dtype) {
    NEED_INIT();
    REQ(head_b && y && ticket && out_pair && dtype == TNN_F32 && M <= 1024 && N >= 16 && N <= 256 && N % 16 == 0 && head_c >= 1 &&
            head_c <= 16,
        "cpu twin: tnn_dense_fwd_head_partials_stats needs every buffer, f32, the head's shapes");
    RECORD(tnn_dense_fwd_head_partials_stats(M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, head_w, head_c, head_z, head_b, y,
                                             ticket, out_pair, exchange, dtype));
    if (int rc = tnn_dense_fwd_head_partials(M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, head_w, head_c, head_z, dtype)) return rc;
    if (exchange == 2) return 0;                      // deferred exchange: no statistics tail (tnn_mlp_head_bwd_tick_xchg follows)
    std::vector<float> logits((size_t)(M * head_c));
    twin_logits_from_partials(M, N, head_c, head_b, head_z, logits.data());
    return tnn_softmax_nll_stats(logits.data(), M, head_c, out_pair, dtype);      // a one-rank exchange is the identity
}
int tnn_dense_fwd_rows_head_stats(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                  const void* bias, int act, int relu_sign, void* C, int64_t ldc, const void* head_w,
                                  int64_t head_c, void* head_z_full, const void* head_b, void* pairs, int dtype) {
    NEED_INIT();
    REQ(head_w && head_z_full && head_b && pairs && dtype == TNN_F32 && M >= 1 && M <= 1024 && N == 128 && head_c == 10 && act == TNN_ACT_RELU,
        "cpu twin: tnn_dense_fwd_rows_head_stats needs every buffer, f32, the head's shapes, a ReLU layer");
    RECORD(tnn_dense_fwd_rows_head_stats(M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, head_w, head_c, head_z_full, head_b,
                                         pairs, dtype));
    if (int rc = tnn_gemm_bias_act(0, 0, M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, dtype)) return rc;
    float* z = (float*)head_z_full;
    for (int64_t r = 0; r < M; ++r)
        for (int64_t c = 0; c < head_c; ++c) {
            float s = 0.f;
            for (int64_t col = 0; col < N; ++col) s += ((const float*)C)[r * ldc + col] * ((const float*)head_w)[col * head_c + c];
            z[r * head_c + c] = s;                                  // the classifier bias is added by the head launch
        }
    for (int64_t p0 = 0; p0 < M; p0 += 16) {                        // one {max, sum-exp} pair per 16-row panel
        const int64_t p1 = p0 + 16 < M ? p0 + 16 : M;
        float mx = -INFINITY;
        for (int64_t r = p0; r < p1; ++r)
            for (int64_t c = 0; c < head_c; ++c) mx = std::max(mx, z[r * head_c + c] + ((const float*)head_b)[c]);
        double se = 0.0;
        for (int64_t r = p0; r < p1; ++r)
            for (int64_t c = 0; c < head_c; ++c) se += std::exp((double)(z[r * head_c + c] + ((const float*)head_b)[c]) - (double)mx);
        ((float*)pairs)[2 * (p0 / 16)] = mx;
        ((float*)pairs)[2 * (p0 / 16) + 1] = (float)se;
    }
    return 0;
}
int tnn_lse_merge(const void*, int, void*, int);
int tnn_dense_fwd_rows_head_stats_merged(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                         const void* bias, int act, int relu_sign, void* C, int64_t ldc, const void* head_w,
                                         int64_t head_c, void* head_z_full, const void* head_b, void* pairs, void* ticket, void* out_pair,
                                         int exchange, int dtype) {
    NEED_INIT();
    REQ(ticket && out_pair, "cpu twin: tnn_dense_fwd_rows_head_stats_merged needs the ticket and out_pair");
    RECORD(tnn_dense_fwd_rows_head_stats_merged(M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, head_w, head_c, head_z_full, head_b,
                                                pairs, ticket, out_pair, exchange, dtype));
    if (int rc = tnn_dense_fwd_rows_head_stats(M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, head_w, head_c, head_z_full, head_b,
                                               pairs, dtype)) return rc;
    if (exchange == 2) return 0;                      // deferred exchange: the panels' pairs only
    return tnn_lse_merge(pairs, (int)((M + 15) / 16), out_pair, dtype);            // a one-rank exchange is the identity
}
int tnn_mlp_head_bwd_tick_ext(int64_t rows, int64_t m_global, int64_t n_in, int64_t nh, int64_t nc, const void* x, const void* w1,
                              const void* a, const void* w, const void* b, const void* y, const void* zpart, const void* pairs,
                              int n_pairs, void* logits, void* dz, void* stats, void* loss, void* dw, void* db, void* dw1, void* db1,
                              void* dx, int dtype, void* pows, double b1, double b2) {
    NEED_INIT();
    REQ(x && w1 && zpart && pairs && logits && dz && dw && db && dw1 && db1 && dx && dtype == TNN_F32 && n_pairs != 0,
        "cpu twin: tnn_mlp_head_bwd_tick_ext needs every buffer, f32");
    RECORD(tnn_mlp_head_bwd_tick_ext(rows, m_global, n_in, nh, nc, x, w1, a, w, b, y, zpart, pairs, n_pairs, logits, dz, stats, loss,
                                     dw, db, dw1, db1, dx, dtype, pows, b1, b2));
    if (n_pairs < 0) {                                              // whole logits without the bias (tnn_dense_fwd_rows_head_stats)
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t c = 0; c < nc; ++c)
                ((float*)logits)[r * nc + c] = ((const float*)zpart)[r * nc + c] + ((const float*)b)[c];
        n_pairs = -n_pairs;
    } else
    twin_logits_from_partials(rows, nh, nc, b, zpart, (float*)logits);
    float merged[2];
    if (int rc = tnn_lse_merge(pairs, n_pairs, merged, dtype)) return rc;
    if (stats) { ((float*)stats)[0] = merged[0]; ((float*)stats)[1] = merged[1]; }
    if (pows) { ((double*)pows)[0] *= b1; ((double*)pows)[1] *= b2; }
    float loss_tmp = 0.f;
    if (int rc = tnn_softmax_nll_fwd_bwd(logits, y, rows, nc, m_global, merged, loss ? loss : &loss_tmp, dz, dtype)) return rc;
    std::vector<float> da((size_t)(rows * nh));
    if (int rc = tnn_dense_bwd(rows, nh, nc, a, dz, w, dw, db, da.data(), a, dtype)) return rc;
    return tnn_dense_bwd(rows, n_in, nh, x, da.data(), w1, dw1, db1, dx, x, dtype);
}
int tnn_mlp_head_bwd_xchg_fits(int64_t, int64_t, int64_t, int64_t, int, int* fits) {
    REQ(fits != nullptr, "tnn_mlp_head_bwd_xchg_fits: fits is NULL");
    *fits = 0;                                           // no peer-to-peer transport on the twin: the step takes the all-gather form
    return 0;
}
int tnn_allgather(const void*, void*, int64_t, int);
int tnn_comm_world(int*, int*);
int tnn_mlp_head_bwd_tick_xchg(int64_t rows, int64_t m_global, int64_t n_in, int64_t nh, int64_t nc, const void* x, const void* w1,
                               const void* a, const void* w, const void* b, const void* y, const void* zpart, const void* shard_pairs,
                               int n_pairs, void* logits, void* dz, void* stats, void* loss, void* dw, void* db, void* dw1, void* db1,
                               void* dx, int dtype, void* pows, double b1, double b2) {
    // the deferred statistics exchange of the HIP library (tnn_p2p.h: XchgCtx) restated with the twin's collective: the shard's
    // pair (from the partial logits, or merged from the row-panel forward's pairs) -> all-gather over the ranks -> the form that
    // takes every rank's pair from memory
    NEED_INIT();
    REQ(x && w1 && zpart && logits && dz && dw && db && dw1 && db1 && dx && dtype == TNN_F32 && n_pairs <= 0 &&
            (n_pairs == 0 || shard_pairs),
        "cpu twin: tnn_mlp_head_bwd_tick_xchg needs every buffer, f32, n_pairs <= 0");
    float mine[2];
    if (n_pairs < 0) {
        if (int rc = tnn_lse_merge(shard_pairs, -n_pairs, mine, dtype)) return rc;
    } else {
        std::vector<float> z((size_t)(rows * nc));
        twin_logits_from_partials(rows, nh, nc, b, zpart, z.data());
        if (int rc = tnn_softmax_nll_stats(z.data(), rows, nc, mine, dtype)) return rc;
    }
    int rank = 0, world = 1;
    if (int rc = tnn_comm_world(&rank, &world)) return rc;
    std::vector<float> all((size_t)world * 2);
    if (world > 1) { if (int rc = tnn_allgather(mine, all.data(), 2, dtype)) return rc; }
    else { all[0] = mine[0]; all[1] = mine[1]; }
    return tnn_mlp_head_bwd_tick_ext(rows, m_global, n_in, nh, nc, x, w1, a, w, b, y, zpart, all.data(), n_pairs < 0 ? -world : world,
                                     logits, dz, stats, loss, dw, db, dw1, db1, dx, dtype, pows, b1, b2);
}
int tnn_dense_fwd_head_partials(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                const void* bias, int act, int relu_sign, void* C, int64_t ldc, const void* hw, int64_t hc,
                                void* hz, int dtype) {
    NEED_INIT();
    REQ(dtype == TNN_F32 && hw && hz && hc >= 1 && hc <= 16, "tnn_dense_fwd_head_partials: f32, head_w, head_z, 1 <= head_c <= 16");
    RECORD(tnn_dense_fwd_head_partials(M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, hw, hc, hz, dtype));
    if (int rc = tnn_gemm_bias_act(0, 0, M, N, K, A, lda, B, ldb, bias, act, relu_sign, C, ldc, dtype)) return rc;
    const int64_t tiles = (N + 15) / 16;
    for (int64_t tn = 0; tn < tiles; ++tn)
        for (int64_t r = 0; r < M; ++r)
            for (int64_t c = 0; c < hc; ++c) {
                float s = 0.f;
                for (int64_t col = tn * 16; col < tn * 16 + 16 && col < N; ++col)
                    s += ((const float*)C)[r * ldc + col] * ((const float*)hw)[col * hc + c];
                ((float*)hz)[(tn * M + r) * hc + c] = s;
            }
    return 0;
}
int tnn_mse_fwd_bwd(const void* pred, const void* y, int64_t n, int64_t mg, void* loss_out, void* dpred, int dtype) {
    NEED_INIT();
    REQ(n > 0 && mg > 0, "tnn_mse_fwd_bwd: empty batch");
    RECORD(tnn_mse_fwd_bwd(pred, y, n, mg, loss_out, dpred, dtype));
    FLOAT_SWITCH(dtype, "tnn_mse_fwd_bwd", {
        double loss = 0;
        for (int64_t i = 0; i < n; ++i) {
            T e = ((const T*)pred)[i] - ((const T*)y)[i];
            loss += (double)e * (double)e;
            if (dpred) ((T*)dpred)[i] = (T)(2.0 / (double)mg) * e;
        }
        if (loss_out) ((T*)loss_out)[0] = (T)(loss / (double)mg);
    });
    return 0;
}
int tnn_mse_fwd_bwd_tick(const void* pred, const void* y, int64_t n, int64_t mg, void* loss_out, void* loss_out2, void* dpred,
                         int dtype, void* pows, double b1, double b2) {
    NEED_INIT();
    REQ(loss_out != nullptr || loss_out2 == nullptr, "tnn_mse_fwd_bwd_tick: loss_out2 needs loss_out");
    RECORD(tnn_mse_fwd_bwd_tick(pred, y, n, mg, loss_out, loss_out2, dpred, dtype, pows, b1, b2));
    if (pows) { ((double*)pows)[0] *= b1; ((double*)pows)[1] *= b2; }
    if (int rc = tnn_mse_fwd_bwd(pred, y, n, mg, loss_out, dpred, dtype)) return rc;
    if (loss_out2) memcpy(loss_out2, loss_out, dtype == TNN_F64 ? 8 : 4);
    return 0;
}
int tnn_sgd(void* p, const void* g, int64_t n, double lr, int dtype) {
    NEED_INIT();
    RECORD(tnn_sgd(p, g, n, lr, dtype));
    FLOAT_SWITCH(dtype, "tnn_sgd", {
        for (int64_t i = 0; i < n; ++i) ((T*)p)[i] = ((T*)p)[i] + (-(T)lr * ((const T*)g)[i]);
    });
    return 0;
}
int tnn_dense_bwd_first_allreduce_adam(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz, void* grads,
                                       int64_t n_reduce, int64_t w_off, int64_t b_off, void* p, void* m, void* v,
                                       int64_t n_params, double lr, double b1, double b2, double eps, const void* pows,
                                       int64_t scalar_index, void* scalar_dst, int dtype) {
    const size_t esz = dtype == TNN_F64 ? 8 : 4;       // no peer-to-peer transport here: the two calls the launch replaces
    if (int rc = tnn_gemm_tn_colsum(n_in, n_out, rows, x, n_in, dz, n_out, (char*)grads + (size_t)w_off * esz, n_out,
                                    (char*)grads + (size_t)b_off * esz, dtype))
        return rc;
    return tnn_allreduce_adam(grads, n_reduce, p, m, v, n_params, lr, b1, b2, eps, const_cast<void*>(pows), 0, dtype,
                              scalar_index, scalar_dst);
}

int tnn_dense_bwd_first_adam(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz, void* dw, void* db,
                             void* p_w, void* m_w, void* v_w, void* p_b, void* m_b, void* v_b, void* flat_p,
                             const void* flat_g, void* flat_m, void* flat_v, int64_t flat_n, double lr, double b1,
                             double b2, double eps, const void* pows, int dtype) {
    std::vector<char> scratch;                       // dw == NULL: the gradient is consumed without being stored
    if (!dw && !g_capturing) { scratch.resize((size_t)(n_in * n_out) * (dtype == TNN_F64 ? 8 : 4)); dw = scratch.data(); }
    REQ(dw != nullptr, "cpu twin: tnn_dense_bwd_first_adam without dw cannot be captured");
    if (int rc = tnn_gemm_tn_colsum(n_in, n_out, rows, x, n_in, dz, n_out, dw, n_out, db, dtype)) return rc;
    void* pw = const_cast<void*>(pows);
    if (int rc = tnn_adam_ex(p_w, dw, m_w, v_w, n_in * n_out, lr, b1, b2, eps, pw, nullptr, dtype, 0, nullptr, nullptr)) return rc;
    if (int rc = tnn_adam_ex(p_b, db, m_b, v_b, n_out, lr, b1, b2, eps, pw, nullptr, dtype, 0, nullptr, nullptr)) return rc;
    if (flat_n > 0)
        return tnn_adam_ex(flat_p, flat_g, flat_m, flat_v, flat_n, lr, b1, b2, eps, pw, nullptr, dtype, 0, nullptr, nullptr);
    return 0;
}
int tnn_optim_step(int kind, void* p, const void* g, void* s1, void* s2, void* step_out, int64_t n, double lr, double a,
                   double b, double eps, int dtype) {
    NEED_INIT();
    REQ(kind >= TNN_OPT_MOMENTUM && kind <= TNN_OPT_ADADELTA, "tnn_optim_step: unknown optimizer");
    REQ(g && s1 && (p || step_out), "tnn_optim_step: g, s1 and one of p / step_out are required");
    REQ(s2 || kind == TNN_OPT_MOMENTUM || kind == TNN_OPT_ADAGRAD, "tnn_optim_step: this optimizer needs s2");
    RECORD(tnn_optim_step(kind, p, g, s1, s2, step_out, n, lr, a, b, eps, dtype));
    FLOAT_SWITCH(dtype, "tnn_optim_step", {
        T* S1 = (T*)s1; T* S2 = (T*)s2;
        for (int64_t i = 0; i < n; ++i) {
            const T gi = ((const T*)g)[i];
            T step;
            if (kind == TNN_OPT_MOMENTUM) {
                S1[i] = (T)a * S1[i] + gi;
                step = -(T)lr * S1[i];
            } else if (kind == TNN_OPT_RMSPROP) {
                S1[i] = S1[i] + ((T)1 - (T)a) * (gi * gi - S1[i]);
                S2[i] = (T)b * S2[i] + (T)lr * gi / (T)sqrt((double)(S1[i] + (T)eps));
                step = -S2[i];
            } else if (kind == TNN_OPT_ADAGRAD) {
                S1[i] = S1[i] + gi * gi;
                step = -((T)lr / (T)sqrt((double)(S1[i] + (T)eps))) * gi;
            } else {
                S1[i] = S1[i] + ((T)1 - (T)a) * (gi * gi - S1[i]);
                const T delta = gi * ((T)sqrt((double)(S2[i] + (T)eps)) / (T)sqrt((double)(S1[i] + (T)eps)));
                step = -(T)lr * delta;
                S2[i] = S2[i] + ((T)1 - (T)a) * (delta * delta - S2[i]);
            }
            if (step_out) ((T*)step_out)[i] = step;
            if (p) ((T*)p)[i] = ((T*)p)[i] + step;
        }
    });
    return 0;
}
int tnn_adam(void* p, const void* g, void* m, void* v, int64_t n, double lr, double b1, double b2, double eps,
             void* pows, void* step_out, int dtype) {
    return tnn_adam_ex(p, g, m, v, n, lr, b1, b2, eps, pows, step_out, dtype, 1, nullptr, nullptr);
}
int tnn_adam_ex(void* p, const void* g, void* m, void* v, int64_t n, double lr, double b1, double b2, double eps,
                void* pows, void* step_out, int dtype, int advance, const void* scalar_src, void* scalar_dst) {
    NEED_INIT();
    REQ(pows, "tnn_adam: pows state is NULL");
    RECORD(tnn_adam_ex(p, g, m, v, n, lr, b1, b2, eps, pows, step_out, dtype, advance, scalar_src, scalar_dst));
    if (scalar_dst) memcpy(scalar_dst, scalar_src, dtype == TNN_F64 ? 8 : 4);
    double* st = (double*)pows;
    double p1 = advance ? st[0] * b1 : st[0], p2 = advance ? st[1] * b2 : st[1];
    FLOAT_SWITCH(dtype, "tnn_adam", {
        T ic1 = (T)(1.0 / (1.0 - p1)), ic2 = (T)(1.0 / (1.0 - p2));
        for (int64_t i = 0; i < n; ++i) {
            T gi = ((const T*)g)[i];
            T mi = ((T*)m)[i], vi = ((T*)v)[i];
            mi = mi + ((T)1 - (T)b1) * (gi - mi);
            vi = vi + ((T)1 - (T)b2) * (gi * gi - vi);
            ((T*)m)[i] = mi;
            ((T*)v)[i] = vi;
            T s = -(T)lr * (mi * ic1) / ((T)sqrt((double)(vi * ic2)) + (T)eps);
            if (step_out) ((T*)step_out)[i] = s;
            else ((T*)p)[i] = ((T*)p)[i] + s;
        }
    });
    st[0] = p1;
    st[1] = p2;
    return 0;
}

// ---- bf16 path: EMULATION of the device's arithmetic contract (configs[4]) so the product's bf16 trainer host code
// (csrc/tnn_mlp.cpp: bucket order, reduce-scatter / sharded Adam / all-gather, loss slot) runs at world > 1 without a
// GPU.  bf16 = the upper 16 bits of an f32, rounded to nearest even on every store (tinynn-autograd_amd/bf16.py
// round_to_bf16 is the same rule); products accumulate in f32.  The summation ORDER inside a dot product differs from
// the MFMA's, so twin-vs-device agreement is to f32 round-off, not bit-for-bit.
typedef uint16_t bf16_t;
static inline bf16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static inline float bf2f(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// C[M,N] = A[M,K] B[N,K]^T (core/ops.py:151,157,160 through the K-contiguous copies), epilogues of tnn_gemm_bf16_nt
int tnn_gemm_bf16_nt(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                     int64_t ldc, int c_dtype, const void* bias, int act, int relu_sign, const void* mask_y, int64_t ldy) {
    NEED_INIT();
    REQ(A && B && C && M > 0 && N > 0 && K > 0, "tnn_gemm_bf16_nt: bad arguments");
    REQ(c_dtype == TNN_BF16 || c_dtype == TNN_F32, "tnn_gemm_bf16_nt: c_dtype %d", c_dtype);
    REQ(!(bias || act) || !mask_y, "tnn_gemm_bf16_nt: bias/activation and mask epilogues are exclusive");
    RECORD(tnn_gemm_bf16_nt(M, N, K, A, lda, B, ldb, C, ldc, c_dtype, bias, act, relu_sign, mask_y, ldy));
    const bf16_t *a = (const bf16_t*)A, *b = (const bf16_t*)B, *y = (const bf16_t*)mask_y;
    std::vector<float> arow((size_t)K);
    for (int64_t i = 0; i < M; ++i) {
        for (int64_t k = 0; k < K; ++k) arow[k] = bf2f(a[i * lda + k]);
        for (int64_t j = 0; j < N; ++j) {
            float acc = 0.f;
            const bf16_t* bj = b + j * ldb;
            for (int64_t k = 0; k < K; ++k) acc += arow[k] * bf2f(bj[k]);
            if (bias || act) {
                if (bias) acc += ((const float*)bias)[j];
                if (act == TNN_ACT_RELU) acc = acc < 0.f ? (relu_sign ? -0.0f : 0.f) : fabsf(acc);
            } else if (y) {
                if (y[i * ldy + j] & 0x8000u) acc = 0.f;
            }
            if (c_dtype == TNN_BF16) ((bf16_t*)C)[i * ldc + j] = f2bf(acc);
            else ((float*)C)[i * ldc + j] = acc;
        }
    }
    return 0;
}
int tnn_transpose_bf16(const void* in, void* out, int64_t rows, int64_t cols) {
    NEED_INIT();
    RECORD(tnn_transpose_bf16(in, out, rows, cols));
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) ((bf16_t*)out)[c * rows + r] = ((const bf16_t*)in)[r * cols + c];
    return 0;
}
int tnn_transpose2_bf16(const void* in1, void* out1, int64_t r1, int64_t c1, const void* in2, void* out2, int64_t r2, int64_t c2) {
    if (int rc = tnn_transpose_bf16(in1, out1, r1, c1)) return rc;
    return tnn_transpose_bf16(in2, out2, r2, c2);
}
int tnn_cast_bf16(const void* in, void* out, int64_t n, int to_bf16) {
    NEED_INIT();
    RECORD(tnn_cast_bf16(in, out, n, to_bf16));
    for (int64_t i = 0; i < n; ++i) {
        if (to_bf16) ((bf16_t*)out)[i] = f2bf(((const float*)in)[i]);
        else ((float*)out)[i] = bf2f(((const bf16_t*)in)[i]);
    }
    return 0;
}
int tnn_colsum_bf16(const void* in, void* out, int64_t rows, int64_t cols) {     // bias gradient, core/ops.py:52-54
    NEED_INIT();
    RECORD(tnn_colsum_bf16(in, out, rows, cols));
    for (int64_t c = 0; c < cols; ++c) {
        double acc = 0.0;
        for (int64_t r = 0; r < rows; ++r) acc += (double)bf2f(((const bf16_t*)in)[r * cols + c]);
        ((float*)out)[c] = (float)acc;
    }
    return 0;
}
int tnn_mse_bf16(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out, void* dpred) {
    NEED_INIT();
    RECORD(tnn_mse_bf16(pred, y, n, m_global, loss_out, dpred));
    const double inv_m = 1.0 / (double)m_global;
    const float two_inv_m = (float)(2.0 * inv_m);
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const float e = bf2f(((const bf16_t*)pred)[i]) - bf2f(((const bf16_t*)y)[i]);
        acc += (double)e * (double)e;
        if (dpred) ((bf16_t*)dpred)[i] = f2bf(two_inv_m * e);
    }
    ((float*)loss_out)[0] = (float)(acc * inv_m);
    return 0;
}
int tnn_mse_bf16_tick(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out, void* loss_out2, void* dpred,
                      void* pows, double b1, double b2) {
    NEED_INIT();
    REQ(loss_out != nullptr || loss_out2 == nullptr, "tnn_mse_bf16_tick: loss_out2 needs loss_out");
    RECORD(tnn_mse_bf16_tick(pred, y, n, m_global, loss_out, loss_out2, dpred, pows, b1, b2));
    if (pows) { ((double*)pows)[0] *= b1; ((double*)pows)[1] *= b2; }
    float tmp = 0.f;
    if (int rc = tnn_mse_bf16(pred, y, n, m_global, loss_out ? loss_out : &tmp, dpred)) return rc;
    if (loss_out2) memcpy(loss_out2, loss_out, 4);
    return 0;
}
// core/optimizer.py:67-79 on the fp32 master copy; G16 = the gradient arrives as bf16 (reduce-scattered wire format)
static void adam_master_rows(bool G16, float* p, const void* g, float* m, float* v, bf16_t* w16, bf16_t* wT16, int64_t rows,
                             int64_t cols, int64_t ldt, double lr, double b1, double b2, double eps, const double* st) {
    const float ic1 = (float)(1.0 / (1.0 - st[0])), ic2 = (float)(1.0 / (1.0 - st[1]));
    const float omb1 = 1.f - (float)b1, omb2 = 1.f - (float)b2, flr = (float)lr, feps = (float)eps;
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) {
            const int64_t i = r * cols + c;
            const float gi = G16 ? bf2f(((const bf16_t*)g)[i]) : ((const float*)g)[i];
            float mi = m[i], vi = v[i];
            mi = mi + omb1 * (gi - mi);
            vi = vi + omb2 * (gi * gi - vi);
            m[i] = mi;
            v[i] = vi;
            const float pi = p[i] + (-flr * (mi * ic1) / (sqrtf(vi * ic2) + feps));
            p[i] = pi;
            if (w16) w16[i] = f2bf(pi);
            if (wT16) wT16[c * ldt + r] = f2bf(pi);
        }
}
int tnn_adam_master_bf16_2d(void* p, const void* g, void* m, void* v, void* w16, void* wT16, int64_t rows, int64_t cols,
                            double lr, double b1, double b2, double eps, void* pows, int advance) {
    NEED_INIT();
    REQ(p && g && m && v && pows, "tnn_adam_master_bf16_2d: NULL argument");
    RECORD(tnn_adam_master_bf16_2d(p, g, m, v, w16, wT16, rows, cols, lr, b1, b2, eps, pows, advance));
    double* st = (double*)pows;
    if (advance) { st[0] *= b1; st[1] *= b2; }
    adam_master_rows(false, (float*)p, g, (float*)m, (float*)v, (bf16_t*)w16, (bf16_t*)wT16, rows, cols, rows, lr, b1, b2, eps, st);
    return 0;
}
int tnn_adam_master_bf16(void* p, const void* g, void* m, void* v, void* w16, int64_t n, double lr, double b1, double b2,
                         double eps, void* pows) {
    return tnn_adam_master_bf16_2d(p, g, m, v, w16, nullptr, 1, n, lr, b1, b2, eps, pows, 1);
}
int tnn_adam_master_g16(void* p, const void* g16, void* m, void* v, void* w16, int64_t n, double lr, double b1, double b2,
                        double eps, const void* pows) {
    NEED_INIT();
    REQ(p && g16 && m && v && w16 && pows, "tnn_adam_master_g16: NULL argument");
    RECORD(tnn_adam_master_g16(p, g16, m, v, w16, n, lr, b1, b2, eps, pows));
    adam_master_rows(true, (float*)p, g16, (float*)m, (float*)v, (bf16_t*)w16, nullptr, 1, n, 1, lr, b1, b2, eps, (const double*)pows);
    return 0;
}
int tnn_bias_bf16_adam(const void* dz, int64_t rows, int64_t cols, void* db, void* p, void* m, void* v, void* w16, double lr,
                       double b1, double b2, double eps, const void* pows) {
    NEED_INIT();
    REQ(dz && db && rows > 0, "tnn_bias_bf16_adam: dz and db are required");
    RECORD(tnn_bias_bf16_adam(dz, rows, cols, db, p, m, v, w16, lr, b1, b2, eps, pows));
    if (int rc = tnn_colsum_bf16(dz, db, rows, cols)) return rc;
    if (p) adam_master_rows(false, (float*)p, db, (float*)m, (float*)v, (bf16_t*)w16, nullptr, 1, cols, 1, lr, b1, b2, eps,
                            (const double*)pows);
    return 0;
}
int tnn_gemm_bf16_nt_adam(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* g_out,
                          void* p, void* m, void* v, void* w16, void* wT16, double lr, double b1, double b2, double eps,
                          const void* pows) {
    NEED_INIT();
    REQ(p && m && v && pows, "tnn_gemm_bf16_nt_adam: NULL argument");
    RECORD(tnn_gemm_bf16_nt_adam(M, N, K, A, lda, B, ldb, g_out, p, m, v, w16, wT16, lr, b1, b2, eps, pows));
    std::vector<float> g((size_t)(M * N));
    if (int rc = tnn_gemm_bf16_nt(M, N, K, A, lda, B, ldb, g.data(), N, TNN_F32, nullptr, TNN_ACT_NONE, 0, nullptr, 0)) return rc;
    if (g_out) memcpy(g_out, g.data(), g.size() * 4);
    adam_master_rows(false, (float*)p, g.data(), (float*)m, (float*)v, (bf16_t*)w16, (bf16_t*)wT16, M, N, M, lr, b1, b2, eps,
                            (const double*)pows);
    return 0;
}
int tnn_bias_bf16_adam_multi(int n_layers, const void* const* dz, int64_t rows, const int64_t* cols, void* const* db, void* const* p,
                             void* const* m, void* const* v, void* const* w16, double lr, double b1, double b2, double eps,
                             const void* pows) {
    NEED_INIT();
    REQ(n_layers >= 1 && n_layers <= 16 && dz && cols && db && rows > 0, "tnn_bias_bf16_adam_multi: bad arguments");
    // (the pointer arrays belong to the caller: copied before a capture stores the call)
    std::vector<const void*> dzv(dz, dz + n_layers);
    std::vector<int64_t> cv(cols, cols + n_layers);
    std::vector<void*> dbv(db, db + n_layers), pv, mv, vv, wv;
    if (p) pv.assign(p, p + n_layers);
    if (m) mv.assign(m, m + n_layers);
    if (v) vv.assign(v, v + n_layers);
    if (w16) wv.assign(w16, w16 + n_layers);
    auto run = [=]() -> int {
        for (int l = 0; l < n_layers; ++l)
            if (int rc = tnn_bias_bf16_adam(dzv[l], rows, cv[l], dbv[l], pv.empty() ? nullptr : pv[l], mv.empty() ? nullptr : mv[l],
                                            vv.empty() ? nullptr : vv[l], wv.empty() ? nullptr : wv[l], lr, b1, b2, eps, pows))
                return rc;
        return 0;
    };
    if (g_capturing) { g_capturing->calls.push_back(run); return 0; }
    return run();
}
int tnn_gemm_bf16_nt_t(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                       const void* bias, int act, int relu_sign, const void* mask_y, int64_t ldy, void* Ct, int64_t ldct) {
    NEED_INIT();
    REQ(Ct != nullptr && ldct >= M, "tnn_gemm_bf16_nt_t: C_t is required with ldct >= M");
    RECORD(tnn_gemm_bf16_nt_t(M, N, K, A, lda, B, ldb, C, ldc, bias, act, relu_sign, mask_y, ldy, Ct, ldct));
    if (int rc = tnn_gemm_bf16_nt(M, N, K, A, lda, B, ldb, C, ldc, TNN_BF16, bias, act, relu_sign, mask_y, ldy)) return rc;
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j < N; ++j) ((bf16_t*)Ct)[j * ldct + i] = ((const bf16_t*)C)[i * ldc + j];
    return 0;
}
int tnn_gemm_bf16_reserve(int64_t, int64_t, int64_t) {
    NEED_INIT();
    return 0;
}
int tnn_mse_bf16_prep(const void* pred, const void* y, int64_t rows, int64_t cols, int64_t m_global, void* loss_out, void* loss_out2,
                      void* dpred, void* dpred_t, const void* x, int64_t x_cols, void* x_t, void* partials, void* ticket, void* pows,
                      double b1, double b2) {
    NEED_INIT();
    REQ(pred && y && rows > 0 && cols > 0 && rows % 64 == 0 && cols % 64 == 0 && partials && ticket && (x == nullptr) == (x_t == nullptr) &&
            (x == nullptr || x_cols % 64 == 0),
        "tnn_mse_bf16_prep: bad arguments");
    REQ(loss_out != nullptr || loss_out2 == nullptr, "tnn_mse_bf16_prep: loss_out2 needs loss_out");
    RECORD(tnn_mse_bf16_prep(pred, y, rows, cols, m_global, loss_out, loss_out2, dpred, dpred_t, x, x_cols, x_t, partials, ticket, pows, b1, b2));
    std::vector<bf16_t> dz((size_t)(rows * cols));
    if (int rc = tnn_mse_bf16_tick(pred, y, rows * cols, m_global, loss_out, loss_out2, dpred ? dpred : dz.data(), pows, b1, b2)) return rc;
    const bf16_t* d = dpred ? (const bf16_t*)dpred : dz.data();
    if (dpred_t)
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t c = 0; c < cols; ++c) ((bf16_t*)dpred_t)[c * rows + r] = d[r * cols + c];
    if (x_t)
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t c = 0; c < x_cols; ++c) ((bf16_t*)x_t)[c * rows + r] = ((const bf16_t*)x)[r * x_cols + c];
    return 0;
}
int tnn_adam_tick(void* pows, double b1, double b2) {
    NEED_INIT();
    REQ(pows != nullptr, "tnn_adam_tick: pows state is NULL");
    RECORD(tnn_adam_tick(pows, b1, b2));
    ((double*)pows)[0] *= b1;
    ((double*)pows)[1] *= b2;
    return 0;
}

// ---- comm: single-process identity by default.  Multi-process CPU tests either use gloo at the Python layer
// (GlooCommunicator) or — to run the product's own C++ data-parallel step (tnn_mlp_step_sharded: bucketing, per-layer
// Adam order, loss slot) at world > 1 without a GPU — install two host callbacks here (tnn_twin_set_collectives, TWIN ONLY,
// not part of include/tnn_hip.h): the twin's "device" memory is host memory, so the test's callbacks run the collective
// over torch.distributed/gloo in place.
static int g_comm = 0;
typedef int (*twin_allreduce_fn)(void* buf, int64_t n, int dtype, int rop);
typedef int (*twin_allgather_fn)(const void* send, void* recv, int64_t n_per_rank, int dtype);
static twin_allreduce_fn g_hook_allreduce = nullptr;
static twin_allgather_fn g_hook_allgather = nullptr;
typedef int (*twin_reduce_scatter_fn)(const void* send, void* recv, int64_t n_per_rank, int dtype);
static twin_reduce_scatter_fn g_hook_reduce_scatter = nullptr;
static int g_hook_rank = 0, g_hook_world = 1;
int tnn_twin_set_collectives(int rank, int world, twin_allreduce_fn ar, twin_allgather_fn ag) {
    REQ(world >= 1 && rank >= 0 && rank < world, "tnn_twin_set_collectives: rank %d / world %d", rank, world);
    g_hook_allreduce = ar;
    g_hook_allgather = ag;
    g_hook_rank = ar ? rank : 0;
    g_hook_world = ar ? world : 1;
    g_comm = ar ? 1 : 0;
    return 0;
}
int tnn_twin_set_reduce_scatter(twin_reduce_scatter_fn rs) { g_hook_reduce_scatter = rs; return 0; }
int tnn_comm_unique_id(void* id) { memset(id, 0, 128); return 0; }
int tnn_comm_init(int rank, int world, const void*) {
    REQ(world == 1 && rank == 0, "cpu twin: tnn_comm supports world size 1 only");
    g_comm = 1;
    return 0;
}
int tnn_comm_destroy(void) { g_comm = 0; return 0; }
int tnn_comm_world(int* r, int* w) { if (r) *r = g_hook_rank; if (w) *w = g_hook_world; return 0; }
int tnn_allreduce(void* buf, int64_t n, int dtype, int rop) {
    REQ(g_comm, "tnn_allreduce: tnn_comm_init() has not been called");
    if (g_hook_allreduce && n > 0) {
        REQ(g_hook_allreduce(buf, n, dtype, rop) == 0, "tnn_allreduce: the test's collective callback failed");
    }
    return 0;
}
int tnn_allreduce_async(void* buf, int64_t n, int dtype, int rop) { return tnn_allreduce(buf, n, dtype, rop); }
int tnn_comm_join(void) { return 0; }
// the twin has one "stream": a chain runs inline
int tnn_comm_chain_begin(void) { return 0; }
int tnn_comm_chain_end(void) { return 0; }
int tnn_reduce_scatter(const void* s, void* r, int64_t n, int dtype) {
    REQ(g_comm, "tnn_reduce_scatter: tnn_comm_init() has not been called");
    if (g_hook_reduce_scatter) {
        REQ(g_hook_reduce_scatter(s, r, n, dtype) == 0, "tnn_reduce_scatter: the test's collective callback failed");
        return 0;
    }
    REQ(g_hook_world == 1, "tnn_reduce_scatter: no reduce-scatter callback installed");
    if (r != s) memmove(r, s, (size_t)n * dsize(dtype));
    return 0;
}
int tnn_comm_wait_oldest(void) { return 0; }
int tnn_allgather(const void* s, void* r, int64_t n, int dtype) {
    REQ(g_comm, "tnn_allgather: tnn_comm_init() has not been called");
    if (g_hook_allgather) {
        REQ(g_hook_allgather(s, r, n, dtype) == 0, "tnn_allgather: the test's collective callback failed");
        return 0;
    }
    memmove(r, s, (size_t)n * dsize(dtype));
    return 0;
}

int tnn_softmax_nll_fused_tick(const void* z, const void* y, int64_t m, int64_t c, int64_t m_global, int sharded,
                               void* stats_out, void* loss_out, void* dz, int dtype, void* pows, double b1, double b2) {
    NEED_INIT();
    REQ(m_global == m, "cpu twin: one-rank group only");
    REQ(((c <= 16) || m * c <= (dtype == TNN_F32 ? 4096 : 2048)) && m <= 1024, "tnn_softmax_nll_fused_tick: does not fit one workgroup");
    RECORD(tnn_softmax_nll_fused_tick(z, y, m, c, m_global, sharded, stats_out, loss_out, dz, dtype, pows, b1, b2));
    if (pows) { ((double*)pows)[0] *= b1; ((double*)pows)[1] *= b2; }
    return tnn_softmax_nll_fused(z, y, m, c, stats_out, loss_out, dz, dtype);
}
int tnn_softmax_nll_fused_sharded(const void* z, const void* y, int64_t m, int64_t c, int64_t m_global, void* stats_out,
                                  void* loss_out, void* dz, int dtype) {
    return tnn_softmax_nll_fused_tick(z, y, m, c, m_global, 1, stats_out, loss_out, dz, dtype, nullptr, 0.0, 0.0);
}
int tnn_allreduce_adam(void* grads, int64_t n_reduce, void* p, void* m, void* v, int64_t n_params, double lr, double b1,
                       double b2, double eps, void* pows, int advance, int dtype, int64_t scalar_index, void* scalar_dst) {
    REQ(n_params > 0 && n_reduce >= n_params, "tnn_allreduce_adam: n_reduce < n_params");
    if (int rc = tnn_allreduce(grads, n_reduce, dtype, TNN_RSUM)) return rc;
    const size_t esz = dtype == TNN_F64 ? 8 : 4;
    return tnn_adam_ex(p, grads, m, v, n_params, lr, b1, b2, eps, pows, nullptr, dtype, advance,
                       scalar_dst ? (const char*)grads + (size_t)scalar_index * esz : nullptr, scalar_dst);
}
// the peer-to-peer transport needs device IPC: the twin only knows the one-rank group
int tnn_p2p_create(int rank, int world, int64_t, void* h) {
    REQ(world == 1 && rank == 0, "cpu twin: tnn_p2p supports world size 1 only");
    memset(h, 0, 64);
    return 0;
}
int tnn_p2p_connect(const void*) { g_comm = 1; return 0; }
int tnn_p2p_enable(int) { return 0; }
int tnn_p2p_tune(int) { return 0; }
int tnn_p2p_xchg_selftest(double, double, void*) { REQ(false, "cpu twin: no peer-to-peer transport"); return 0; }
int tnn_p2p_set_bulk_bytes(int64_t) { return 0; }          // (no peer-to-peer transport on the twin: tnn_p2p_create refuses)
int tnn_p2p_status(int* c, int* e, int* d) {
    // with the collective callbacks installed there is no peer-to-peer transport: the step takes the RCCL-shaped path
    const int on = g_comm && !g_hook_allreduce;
    if (c) *c = on; if (e) *e = on; if (d) *d = 0;
    return 0;
}
int tnn_p2p_poll_failed(int* f) { if (f) *f = 0; return 0; }
int tnn_p2p_guard_updates(int) { return 0; }
int tnn_p2p_debug(int* w) { if (w) memset(w, 0, 16 * sizeof(int)); return 0; }
int tnn_p2p_destroy(void) { g_comm = 0; return 0; }

}  // extern "C"

// the product's own trainer host code on top of the primitives above
#include "../../tinynn-autograd_amd/csrc/tnn_mlp.cpp"
