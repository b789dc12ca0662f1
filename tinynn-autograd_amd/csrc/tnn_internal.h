// Internal helpers shared by the HIP translation units of libtnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tnn_hip.h"

enum { TNN_FAULT_SPLITK_HANDOFF = 1 };

namespace tnn {
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
hipStream_t stream();          // the one library stream (valid after tnn_init) — or the override below while one is set
// tnn_comm_chain_begin/_end: launches of the calling thread go to `s` (the communication stream) until reset with nullptr
void set_stream_override(hipStream_t s);
bool initialised();
int num_cus();                 // 256 on MI355X
// Device word the optimizer-update kernels look at before touching anything: non-zero -> the launch is a no-op
// (parameters, optimizer state and beta powers keep their contents).  nullptr = no guard.  Set around a data-parallel
// step to the peer-to-peer transport's sticky `dead` word (tnn_p2p_guard_updates), so an update that would consume a
// discarded collective is discarded too, whichever kernel applies it.
// Process-wide sticky fault word in host-pinned, device-visible memory: a kernel whose bounded in-launch wait runs out
// stores a TNN_FAULT_* code there (system scope) instead of consuming data that never arrived; tnn_stream_sync and
// tnn_memcpy_d2h report it (rc 1 + tnn_last_error) from then on.
int* fault_word();
int check_fault(const char* where);
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

// Debug build only (make trace: -DTNN_STEP_TRACE): 100 MHz wall-clock stamps of the headline step's four launches, one row of
// four words per workgroup — [0] entry, [1] every operand of the product in registers (s_waitcnt vmcnt(0) behind the last
// load), [2] last MFMA issued and the cross-wave reduction read back, [3] last store acknowledged.  Kernel ids: 0 fwd0, 1 fwd1
// + partial logits, 2 head + hidden backward, 3 first-layer backward + Adam.  tools/probes/step_stamps.py reads them.
#ifdef TNN_STEP_TRACE
#define TNN_STEP_STAMP(buf, kid, slot)                                                                              \
    do {                                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x < 1024) (buf)[((kid) * 1024 + blockIdx.x) * 4 + (slot)] = wall_clock64(); \
    } while (0)
#define TNN_STEP_STAMP_ACKED(buf, kid, slot)                                                                        \
    do {                                                                                                            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
        TNN_STEP_STAMP(buf, kid, slot);                                                                             \
    } while (0)
#else
#define TNN_STEP_STAMP(buf, kid, slot) do { } while (0)
#define TNN_STEP_STAMP_ACKED(buf, kid, slot) do { } while (0)
#endif

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
// ---- DPP reductions: the same wave-wide sums / maxima in ~6 VALU steps per 32-bit half instead of six ds_bpermute
// round trips through the LDS crossbar (a 64-bit __shfl_xor tree costs ~700 cycles, measured; these ~100).  Steps:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror (every lane of a 16-lane row then holds the row's
// result), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's result, read back
// with v_readlane.  ALL 64 lanes must be active (call from wave-uniform control flow); `identity` fills the lanes a
// row-masked step leaves out.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double identity, double v) {
    const long long iv = __double_as_longlong(v), ii = __double_as_longlong(identity);
    const int lo = __builtin_amdgcn_update_dpp((int)ii, (int)iv, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(ii >> 32), (int)(iv >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float read_lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ double read_lane63(double v) {
    const long long iv = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)iv, 63), hi = __builtin_amdgcn_readlane((int)(iv >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
    v += dpp_move<0xB1, 0xf>(T(0), v);
    v += dpp_move<0x4E, 0xf>(T(0), v);
    v += dpp_move<0x141, 0xf>(T(0), v);
    v += dpp_move<0x140, 0xf>(T(0), v);
    v += dpp_move<0x142, 0xa>(T(0), v);
    v += dpp_move<0x143, 0xc>(T(0), v);
    return read_lane63(v);
}
template <typename T>
__device__ __forceinline__ T wave_max_dpp(T v) {
    const T lowest = (T)-INFINITY;
    T w;
    w = dpp_move<0xB1, 0xf>(lowest, v); v = w > v ? w : v;
    w = dpp_move<0x4E, 0xf>(lowest, v); v = w > v ? w : v;
    w = dpp_move<0x141, 0xf>(lowest, v); v = w > v ? w : v;
    w = dpp_move<0x140, 0xf>(lowest, v); v = w > v ? w : v;
    w = dpp_move<0x142, 0xa>(lowest, v); v = w > v ? w : v;
    w = dpp_move<0x143, 0xc>(lowest, v); v = w > v ? w : v;
    return read_lane63(v);
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
