// Internal helpers shared by the HIP translation units of libtnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tnn_hip.h"

namespace tnn {
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
hipStream_t stream();          // the one library stream (valid after tnn_init)
bool initialised();
int num_cus();                 // 256 on MI355X
// Device word the optimizer-update kernels look at before touching anything: non-zero -> the launch is a no-op
// (parameters, optimizer state and beta powers keep their contents).  nullptr = no guard.  Set around a data-parallel
// step to the peer-to-peer transport's sticky `dead` word (tnn_p2p_guard_updates), so an update that would consume a
// discarded collective is discarded too, whichever kernel applies it.
const int* update_guard();
void set_update_guard(const int* device_word);
}  // namespace tnn

#define TNN_GUARD_RETURN(guard)                                                                              \
    do {                                                                                                     \
        if ((guard) != nullptr && __hip_atomic_load((guard), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return; \
    } while (0)

#define TNN_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            tnn::set_error("%s -> %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,  \
                           __LINE__);                                                    \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

#define TNN_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            tnn::set_error(__VA_ARGS__);    \
            return 2;                       \
        }                                   \
    } while (0)

#define TNN_NEED_INIT() TNN_REQUIRE(tnn::initialised(), "tnn_init() has not been called")
#define TNN_LAUNCH_OK() TNN_CHECK_HIP(hipGetLastError())

namespace tnn {

constexpr int kWave = 64;  // CDNA wavefront

// grid for an HBM-bound streaming kernel: enough 256-thread blocks to cover n items, capped at
// 8 blocks per CU (2048 on MI355X) with a grid-stride loop for the rest.
inline unsigned stream_grid(int64_t work_items, int threads = 256) {
    int64_t b = (work_items + threads - 1) / threads;
    int64_t cap = (int64_t)num_cus() * 8;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

}  // namespace tnn
