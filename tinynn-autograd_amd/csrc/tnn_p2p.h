// Internal interface between tnn_comm.hip (the tnn_allreduce / tnn_allgather front), the xGMI peer-to-peer
// transport in tnn_p2p.hip, and kernels elsewhere that embed an exchange step (the sharded softmax head in
// tnn_fused.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "tnn_internal.h"

namespace tnn {
namespace p2p {

constexpr int MAXW = 16;          // ranks
constexpr int MAXB = 128;         // workgroups of the all-reduce kernel
constexpr int AG_BYTES = 256;     // per-rank payload limit of the small all-gather
constexpr int THREADS = 512;

// Flag words are polled straight from DRAM (uncached).  With every workgroup's words packed into one 4 KB page all
// pollers hit one HBM channel and the flag stores queue behind them (measured: 64 workgroups cost +7 us per
// all-reduce over one); each workgroup's words therefore get a row of their own, FLAG_ROW bytes apart.
constexpr int FLAG_ROW = 4096;
struct Header {                                   // start of every rank's uncached region
    uint8_t flag[2][MAXB][FLAG_ROW];              // [barrier][block] -> uint32_t[MAXW] indexed by source rank
    uint32_t ag_flag[MAXW];                       // [source rank]
    uint32_t ag_slot[2][MAXW][AG_BYTES / 4];      // [epoch parity][source rank][word]
    uint64_t ll[2][MAXW][2];                      // [epoch parity][source rank][word]: (tag << 32) | payload, see ll_exchange2
};
constexpr size_t HEADER_BYTES = (sizeof(Header) + 4095) / 4096 * 4096;

struct Peers {
    char* base[MAXW];                             // every rank's region in THIS process' address space
    int rank, world;
    int64_t slice_cap;                            // floats per slice the regions were sized for
    int* dead_host;                               // host-pinned mirror of the sticky `dead` word (device-visible address):
                                                  // written once when a barrier times out, read by the host WITHOUT a sync
};

// A barrier timed out: the sticky device word stops every later wait, the host mirror lets the next host-side call fail
// loudly (tnn_p2p.hip: p2p_failed) instead of running on partial sums.
// `why` (non-zero) says which wait gave up — 1: a flag barrier of a collective, 2: the tagged {max, sum-exp} exchange;
// tnn_p2p_status reports the word as it is.
// The host mirror is 16 ints: [0] the word, [1..4] what the FIRST wait that gave up was looking at (expected value, last
// value seen, peer / workgroup, a wait-specific detail) — tnn_p2p_debug reads them without a stream sync.
__device__ __forceinline__ void mark_dead(const Peers& p, int* dead, int why = 1, uint32_t expected = 0, uint32_t seen = 0,
                                          uint32_t who = 0, uint32_t detail = 0) {
    if (__hip_atomic_exchange(dead, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        p.dead_host[1] = (int)expected; p.dead_host[2] = (int)seen; p.dead_host[3] = (int)who; p.dead_host[4] = (int)detail;
        __hip_atomic_store(p.dead_host, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct LaunchCtx {                                // what a kernel embedding an exchange needs
    Peers peers;
    uint32_t* ag_epoch;                           // epoch of the small all-gather / tagged-exchange slots (device); only
                                                  // single-workgroup exchanges use it, so the exchanging kernel advances it itself
    int* dead;
    int64_t timeout_ticks;
};

// Signal `val` to every peer's word [.. + rank] and wait until every peer's signal arrived in mine.
// Everything that crosses a device boundary lives in UNCACHED memory (stores go straight to the fabric, loads come
// from memory), so no L2 write-back / invalidate is needed — and none is issued: a system-scope release fence is a
// whole-L2 `buffer_wbl2` per workgroup (measured: 64 workgroups -> +12 us per all-reduce).  What IS needed is
// order: each thread waits until its own stores were acknowledged (s_waitcnt vmcnt(0)), the workgroup meets, and only
// then the flag words go out as relaxed system-scope stores; the poll is a relaxed system-scope load (cache-bypassing).
// An agent-scope acquire after the meeting (`buffer_inv sc1`) was measured too: +6 us per all-reduce at 64
// workgroups.  Instead every load of peer-written data carries sc0 sc1 itself (load_sys below), so it cannot be served
// from a vector-L1 or L2 line whatever the page's cache policy turns out to be.
// Returns true (uniformly over the workgroup) when every peer's signal arrived; false when this barrier timed out or
// the transport was already dead — the caller must then NOT consume peer data (no reduction, no copy-out, no optimizer
// update: buffers and parameters stay untouched).
__device__ __forceinline__ bool exchange_flags(const Peers& p, size_t word_offset_bytes, uint32_t val, int* dead,
                                               int64_t timeout_ticks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int t = threadIdx.x;
    int failed = 0;
    if (t < p.world) {
        uint32_t* theirs = reinterpret_cast<uint32_t*>(p.base[t] + word_offset_bytes) + p.rank;
        __hip_atomic_store(theirs, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        uint32_t* mine = reinterpret_cast<uint32_t*>(p.base[p.rank] + word_offset_bytes) + t;
        if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            uint64_t t0 = 0;
            uint32_t polls = 0;
            while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - val) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if ((++polls & 63u) == 0) {                       // look at the clock now and then
                    const uint64_t now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    if ((int64_t)(now - t0) > timeout_ticks) {
                        mark_dead(p, dead, 1, val, __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), (uint32_t)t,
                                  (uint32_t)(word_offset_bytes / FLAG_ROW));
                        failed = 1;
                        break;
                    }
                }
            }
        } else {
            failed = 1;
        }
    }
    failed = __syncthreads_or(failed);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return failed == 0;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-B load that bypasses the vector L1 and L2 (system-scope bits on the instruction).  The compiler does not see
// the outstanding load: issue a batch, then loads_landed() on the batch before the first use.
__device__ __forceinline__ void load_sys(f32x4& v, const float* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void load_sys(uint32_t& v, const uint32_t* ptr) {
    asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=&v"(v) : "v"(ptr) : "memory");
}
// Stores into a peer's region carry the same bits (write-through to the fabric at system scope), so they do not
// depend on how the importing process happened to map the peer's pages; exchange_flags() waits for their acks.
// (s_nop: wait states of the ">64-bit VMEM store followed by a VALU write of its data registers" hazard, which the
// compiler cannot insert around inline asm — see ll_store16 in tnn_p2p.hip)
__device__ __forceinline__ void store_sys(float* ptr, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ void store_sys(uint32_t* ptr, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ void load_sys(uint64_t& v, const uint64_t* ptr) {
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void store_sys(uint64_t* ptr, uint64_t v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(ptr), "v"(v) : "memory");
}
template <typename T, int N>
__device__ __forceinline__ void loads_landed(T (&v)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));      // uses of v[k] stay behind the wait
}

// Two floats per rank to every rank, "low-latency" style: payload and a tag travel in ONE naturally aligned 8-byte
// store (a single fabric transaction), so the receiver simply polls the word until the tag is the one it expects —
// no acknowledgement wait on the sender, no separate flag, no second read: one link latency end to end instead of
// three.  Threads 0 .. 2W-1 take part (thread t: peer t >> 1, word t & 1); returns this thread's received float.
// Slots are double-buffered on the epoch's parity like the all-gather's.
__device__ __forceinline__ float ll_exchange2(const Peers& p, uint32_t epoch, float mine, int* dead, int64_t timeout_ticks,
                                              size_t slots = offsetof(Header, ll)) {
    const int t = threadIdx.x, q = t >> 1, idx = t & 1;
    const uint32_t tag = epoch + 1;
    const size_t base = slots + (size_t)(epoch & 1) * MAXW * 16;
    store_sys(reinterpret_cast<uint64_t*>(p.base[q] + base + (size_t)p.rank * 16) + idx,
              ((uint64_t)tag << 32) | (uint64_t)__float_as_uint(mine));
    const uint64_t* src = reinterpret_cast<const uint64_t*>(p.base[p.rank] + base + (size_t)q * 16) + idx;
    uint64_t v[1] = {0};
    if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return 0.f;
    uint64_t t0 = 0;
    uint32_t polls = 0;
    for (;;) {
        load_sys(v[0], src);
        loads_landed(v);
        if ((uint32_t)(v[0] >> 32) == tag) break;
        if ((++polls & 63u) == 0) {
            const uint64_t now = wall_clock64();
            if (t0 == 0) t0 = now;
            if ((int64_t)(now - t0) > timeout_ticks) {
                mark_dead(p, dead, 2, tag, (uint32_t)(v[0] >> 32), (uint32_t)q, epoch & 1);
                break;
            }
        }
    }
    return __uint_as_float((uint32_t)v[0]);
}

}  // namespace p2p

bool p2p_world(int* rank, int* world);                       // false when no peer group exists
bool p2p_failed();                                           // a peer barrier timed out (host mirror, no stream sync)
int p2p_refuse_if_failed(const char* who);                   // 0, or 3 + tnn_last_error() once the transport is dead
bool p2p_can_allreduce(int64_t n, int dtype, int rop);       // enabled, f32 SUM, fits the mapped regions
int p2p_allreduce(float* buf, int64_t n);
// all-reduce with Adam applied in the kernel's last stage (pows already advanced); buf[scalar_index] -> *scalar_dst
int p2p_allreduce_adam(float* buf, int64_t n, float* p, float* m, float* v, int64_t n_params, double lr, double b1,
                       double b2, double eps, const double* pows, int64_t scalar_index, float* scalar_dst);
bool p2p_can_allgather(int64_t n_per_rank, int dtype);       // enabled, <= 256 B per rank
int p2p_allgather(const void* send, void* recv, int64_t n_per_rank, int dtype);
bool p2p_launch_ctx(p2p::LaunchCtx* ctx);                    // false when the transport is not enabled
}  // namespace tnn
