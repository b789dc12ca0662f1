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
constexpr int GATE_STRIDE = 64;  // uint32 words between two gate counters (a 256-B line of their own each)

// Flag words are polled straight from DRAM (uncached).  With every workgroup's words packed into one 4 KB page all
// pollers hit one HBM channel and the flag stores queue behind them (measured: 64 workgroups cost +7 us per
// all-reduce over one); each workgroup's words therefore get a row of their own, FLAG_ROW bytes apart.
constexpr int FLAG_ROW = 4096;
struct Header {                                   // start of every rank's uncached region
    uint8_t flag[2][MAXB][FLAG_ROW];              // [barrier][block] -> uint32_t[MAXW] indexed by source rank
    uint32_t ag_flag[MAXW];                       // [source rank]
    uint32_t ag_slot[2][MAXW][AG_BYTES / 4];      // [epoch parity][source rank][word]
    uint64_t ll[2][MAXW][2];                      // [epoch parity][source rank][word]: (tag << 32) | payload, see ll_exchange2
    int32_t devid[4];                             // PCI {domain, bus, device} of the owner's GPU + 1 (written before the region is
                                                  // exported): how a rank learns which peers share ITS device (p2p_ranks_on_my_device)
};
constexpr size_t HEADER_BYTES = (sizeof(Header) + 4095) / 4096 * 4096;

struct Peers {
    char* base[MAXW];                             // every rank's region in THIS process' address space
    int rank, world;
    int64_t slice_cap;                            // floats per slice the regions were sized for
    int* dead_host;                               // host-pinned mirror of the sticky `dead` word (device-visible address):
                                                  // written once when a barrier times out, read by the host WITHOUT a sync
    int poll_gap, poll_first;                     // tagged polls: s_sleep units (64 clocks) between two polls of a slot group, and
                                                  // before the FIRST poll of stage (B) (TNN_P2P_POLL_GAP / TNN_P2P_POLL_FIRST)
};
__device__ __forceinline__ void poll_pause(int units) {          // s_sleep takes an immediate: pauses of 1 .. 127 units
    for (; units >= 8; units -= 8) __builtin_amdgcn_s_sleep(8);
    for (; units > 0; --units) __builtin_amdgcn_s_sleep(1);
}

// A barrier timed out: the sticky device word stops every later wait, the host mirror lets the next host-side call fail
// loudly (tnn_p2p.hip: p2p_failed) instead of running on partial sums.
// `why` (non-zero) says which wait gave up — 1: a flag barrier / tagged poll of a collective, 2: the tagged {max, sum-exp} exchange,
// 3: a polling workgroup's gate (producer workgroups of its OWN launch that never arrived: allreduce_body);
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
    uint32_t* ar_epoch;                           // [ar_grid] launch counts of the all-reduce workgroups = the tags of its slots (all
                                                  // equal between launches); a kernel that embeds the all-reduce keeps to the same
                                                  // ar_grid polling workgroups, so the entries stay equal whichever kind runs
    unsigned* ar_gate;                            // [MAXB][GATE_STRIDE] arrival counters of such a kernel's producer workgroups
    int ar_grid;
    uint32_t* xchg_seq;                           // launch sequence of the DEFERRED statistics exchange (XchgCtx::seq below)
};

// ---- deferred exchange of the shards' {max, sum-exp} pairs (core/losses.py:26-27 couples the shards) --------------------
// The merged head + hidden-backward launch of a data-parallel step does the exchange ITSELF: every workgroup reduces the
// shard's pair from the logits as the single-GPU launch does, ONE workgroup pushes it to every peer (tagged 16-byte stores,
// one per polling row and peer), and every workgroup polls its row of its own region for the peers' pairs and merges them in
// rank order.  The forward launch in front then has no statistics tail at all (no acknowledgement wait, arrival ticket,
// system-scope re-read of the partial logits, no exchange: 3.2 us of an 8.4 us launch at world 1), and the link latency
// overlaps the head launch's own start-up instead of extending the forward launch.  Nobody waits on a workgroup of its OWN
// launch except through data it needs anyway, and a sender never waits at all — ranks that share a GPU cannot deadlock.
//   tag   = *seq, a launch sequence advanced by the FORWARD launch of the same step (one thread of it; kernel boundary in
//           between), so every workgroup of the head launch reads the same value however late it starts;
//   slots = Header::flag[0][row] (a 4096-B row per polling row): [parity][source rank] x 16 B = {M, tag, S, tag}.
// Reuse: a rank sends step k + 1's pair after its step-k all-reduce, which needs every rank's step-k gradients, which are
// produced behind every rank's step-k head launch — so the slots of step k have been read everywhere.  A reader that finds
// any other tag keeps polling until the bounded wait gives up (sticky `dead` word): wrong statistics are never consumed.
constexpr int XCHG_ROWS = 32;
struct XchgCtx {                                  // lives in DEVICE memory (tnn_p2p.hip keeps it current), kernels take a pointer
    Peers peers;
    uint32_t* seq;
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
    if (q == p.rank) return mine;                 // this rank's own pair never travels (at world 1 nothing does)
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


constexpr int UNROLL = 4;          // independent 16-B accesses in flight per thread and loop trip

__device__ __forceinline__ f32x4 load_guarded(const float* buf, int64_t i, int64_t n) {
    if (i + 4 <= n) return *reinterpret_cast<const f32x4*>(buf + i);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k)
        if (i + k < n) v[k] = buf[i + k];
    return v;
}
__device__ __forceinline__ void store_guarded(float* buf, int64_t i, int64_t n, f32x4 v) {
    if (i + 4 <= n) {
        *reinterpret_cast<f32x4*>(buf + i) = v;
        return;
    }
    for (int k = 0; k < 4; ++k)
        if (i + k < n) buf[i + k] = v[k];
}

// Optional optimizer tail: stage (C) holds the reduced gradient in registers anyway, so Adam
// (core/optimizer.py:67-79, the maths of adam_kernel in tnn_fused.hip) is applied right there — no second launch and
// no second pass over the gradient.  pows must already hold b1^t, b2^t of THIS step.
struct AdamTail {
    float* p;
    float* m;
    float* v;
    int64_t n_params;              // elements [0, n_params) of buf are gradients of p; the rest is only reduced
    float lr, b1, b2, eps;
    const double* pows;
    int64_t scalar_index;          // buf[scalar_index] is also written to *scalar_dst (e.g. the loss); -1 = none
    float* scalar_dst;
};

// ---- tagged ("low-latency") slots for the bulk data: every 4-byte payload word travels next to a 4-byte tag in ONE
// naturally aligned 8-byte half of a 16-byte store, so the receiver polls the DATA itself until every tag is the one it
// expects.  No store-acknowledgement wait, no flag exchange, no second read per stage: a stage costs one fabric latency
// end to end where the flag-barrier version of rounds 1-2 paid three dependent ones (ack wait, flag store -> poll, data
// load).  Wire efficiency is 50 % — irrelevant at 0.94 MB.  A float4 element i of a slice occupies 32 bytes:
//     {p0, tag, p1, tag | p2, tag, p3, tag}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ll_load16(u32x4& v, const char* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(ptr) : "memory");
}
// s_nop: a VMEM store of more than 64 bits reads its data VGPRs for a few cycles after issue; the compiler keeps VALU
// writes away from its OWN stores (a gfx9 hazard it knows) but cannot see through inline asm — without the wait states
// the v_movs that assemble the NEXT slot overwrote word 0 of this one in some lanes (measured: lanes 12-15 of each row)
__device__ __forceinline__ void ll_store16(char* ptr, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
}
// slot of (source rank src, float4 element i) inside a region: recv half (stage A -> B) or out half (stage B -> C)
__device__ __forceinline__ size_t ll_recv_off(const Peers& p, int src, int64_t i) {
    return HEADER_BYTES + (size_t)((int64_t)src * (p.slice_cap / 4) + i) * 32;
}
__device__ __forceinline__ size_t ll_out_off(const Peers& p, int src, int64_t i) {
    return HEADER_BYTES + (size_t)((int64_t)(p.world + src) * (p.slice_cap / 4) + i) * 32;
}
__device__ __forceinline__ void ll_send(char* dst, f32x4 v, uint32_t tag) {
    ll_store16(dst, u32x4{__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag});
    ll_store16(dst + 16, u32x4{__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag});
}
// Poll N slots until every live one carries `tag` in all four tag words; false = the transport is (now) dead.
template <int N>
__device__ __forceinline__ bool ll_poll(const char* const (&src)[N], const bool (&live)[N], f32x4 (&out)[N], uint32_t tag,
                                        const Peers& p, int* dead, int64_t timeout_ticks) {
    uint64_t t0 = 0;
    uint32_t polls = 0;
    for (;;) {
        u32x4 lo[N], hi[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (live[k]) { ll_load16(lo[k], src[k]); ll_load16(hi[k], src[k] + 16); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
        uint32_t seen = tag;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (!live[k]) continue;
            asm volatile("" : "+v"(lo[k]), "+v"(hi[k]));          // uses stay behind the wait
            const bool ready = lo[k][1] == tag && lo[k][3] == tag && hi[k][1] == tag && hi[k][3] == tag;
            if (!ready) { ok = false; seen = lo[k][1]; }
            out[k] = f32x4{__uint_as_float(lo[k][0]), __uint_as_float(lo[k][2]), __uint_as_float(hi[k][0]), __uint_as_float(hi[k][2])};
        }
        if (ok) return true;
        poll_pause(p.poll_gap);
        if ((++polls & 63u) == 0) {
            // (the sticky word every 64th poll, ~0.1 ms: looked at after EVERY poll it put an agent-scope round trip between two polls)
            if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            const uint64_t now = wall_clock64();
            if (t0 == 0) t0 = now;
            if ((int64_t)(now - t0) > timeout_ticks) {
                mark_dead(p, dead, 1, tag, seen, blockIdx.x, threadIdx.x);
                return false;
            }
        }
    }
}


// What a thread of an all-reduce does with ITS elements of every slice, i_k = first + k stride (k = 0 .. while i_k < slice / 4):
// stages (A), (B), (C) of p2p_allreduce_kernel (tnn_p2p.hip has the protocol).  Also the tail of kernels that PRODUCE part of
// the buffer themselves and push it straight into the owners' recv slots (tnn_gemm.hip: dense_bwd0_allreduce_adam_kernel) —
// `skip` names the element ranges of buf that stage (A) must leave to them.
struct SkipRanges {
    int64_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;      // [lo, hi) element ranges of buf, multiples of 4
#ifdef TNN_AR_TRACE
    unsigned long long* trace = nullptr;             // debug build: [4] wall-clock stamps of this workgroup (start, A, B, C)
#endif
    __device__ __forceinline__ bool has(int64_t e) const { return (e >= lo0 && e < hi0) || (e >= lo1 && e < hi1); }
};
// UN: independent 16-B accesses in flight per thread and loop trip in stages (A) and (C)
template <bool ADAM, int UN = UNROLL>
__device__ __forceinline__ void allreduce_body(const Peers& p, float* __restrict__ buf, const int64_t n, const int64_t slice,
                                               const uint32_t tag, int* dead, const int64_t timeout_ticks, const AdamTail& t,
                                               const int64_t first, const int64_t stride, const SkipRanges skip,
                                               const unsigned* gate = nullptr, const unsigned gate_count = 0) {
    const int W = p.world, r = p.rank;
    const int64_t s4 = slice / 4;
    const int cnt = first < s4 ? (int)((s4 - first + stride - 1) / stride) : 0;
    const int items = cnt * W;                            // (element k, slice) pairs of this thread
    // (the sticky word is looked at once, and its load travels WITH the first batch of gradient loads: as the loop's entry condition
    // it was a memory round trip of its own in front of them)
    const int dead_at_entry = __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;

    // (A) slice order starts at my right-hand neighbour so the links fill evenly.  Loads first (L2 hits: the gradients
    // were just written), then the posted stores.
    for (int j0 = 0; ok && j0 < items; j0 += UN) {
        f32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u;
            if (j < items) {
                const int q = (r + 1 + j % W) % W;
                v[u] = load_guarded(buf, (int64_t)q * slice + 4 * (first + (j / W) * stride), n);
            }
        }
        if (dead_at_entry != 0) { ok = false; break; }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u;
            if (j < items) {
                const int q = (r + 1 + j % W) % W;
                const int64_t i = first + (j / W) * stride;
                if (!skip.has((int64_t)q * slice + 4 * i)) ll_send(p.base[q] + ll_recv_off(p, r, i), v[u], tag);
            }
        }
    }

    ok = ok && dead_at_entry == 0;                          // (threads without an element of their own)
#ifdef TNN_AR_TRACE
    if (skip.trace && threadIdx.x == 0) skip.trace[1] = wall_clock64();
#endif
    // What becomes of a finished float4 of the sum (element offset i0 of buf; n = nothing): the caller's buffer, and with ADAM
    // the optimizer update on the spot (pm / mm / vm: the parameter and moments, requested before the wait that produced g).
    float ic1 = 0.f, ic2 = 0.f, omb1 = 0.f, omb2 = 0.f;
    if constexpr (ADAM) {
        ic1 = (float)(1.0 / (1.0 - t.pows[0]));
        ic2 = (float)(1.0 / (1.0 - t.pows[1]));
        omb1 = 1.f - t.b1;
        omb2 = 1.f - t.b2;
    }
    auto adam1 = [&](float g, float& mi, float& vi, float& pi) {
        mi = mi + omb1 * (g - mi);
        vi = vi + omb2 * (g * g - vi);
        const float mh = mi * ic1, vh = vi * ic2;
        pi = pi + (-t.lr * mh / (sqrtf(vh) + t.eps));
    };
    auto consume = [&](const int64_t i0, const f32x4 g, f32x4 pm, f32x4 mm, f32x4 vm) {
        if (i0 >= n) return;
        if constexpr (ADAM) {
            if (i0 + 4 <= t.n_params) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float mi = mm[k], vi = vm[k], pi = pm[k];
                    adam1(g[k], mi, vi, pi);
                    mm[k] = mi; vm[k] = vi; pm[k] = pi;
                }
                *reinterpret_cast<f32x4*>(t.m + i0) = mm;
                *reinterpret_cast<f32x4*>(t.v + i0) = vm;
                *reinterpret_cast<f32x4*>(t.p + i0) = pm;
                *reinterpret_cast<f32x4*>(buf + i0) = g;
            } else {                                         // the arena's ragged end and the slots behind it
                for (int k = 0; k < 4; ++k) {
                    const int64_t i = i0 + k;
                    if (i >= n) break;
                    if (i < t.n_params) {
                        float mi = t.m[i], vi = t.v[i], pi = t.p[i];
                        adam1(g[k], mi, vi, pi);
                        t.m[i] = mi; t.v[i] = vi; t.p[i] = pi;
                    }
                    buf[i] = g[k];
                    if (i == t.scalar_index) *t.scalar_dst = g[k];
                }
            }
        } else {
            store_guarded(buf, i0, n, g);
        }
    };

    // (B) reduce my slice in rank order and broadcast the result.  The (element, source) pairs of this thread are walked in
    // order, GB slots requested together (eight at once spilled registers): a thread with several elements and few
    // sources polls them side by side instead of paying one memory round trip per element; the sum of an element still
    // runs over the sources in rank order.  The finished sum goes to every PEER's out slots — and is consumed right here for
    // this rank itself (round 6): the owner of a slice has the sum in registers, so its own copy neither travels through an
    // out slot nor waits for stage (C)'s poll; at world 1 stage (C) disappears, at world W the owner's share of the optimizer
    // pass (1 / W of it) runs while the other owners' sums are still on the links.
    {
        constexpr int GB = ADAM ? (UN < 4 ? UN : 4) : 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (gate != nullptr) poll_pause(p.poll_first);        // producers of THIS launch: nothing can have arrived before their product
        for (int j0 = 0; ok && j0 < items; j0 += GB) {
            const char* src[GB];
            bool live[GB];
            f32x4 part[GB];
            f32x4 pm[GB] = {}, mm[GB] = {}, vm[GB] = {};
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int j = j0 + u;
                live[u] = j < items;
                const int64_t i = first + (live[u] ? j / W : 0) * stride;
                src[u] = p.base[r] + ll_recv_off(p, live[u] ? j % W : 0, i);
                if constexpr (ADAM) {
                    // the parameter / moment loads of an element this group finishes do not depend on the peers: issued under
                    // the same wait
                    const int64_t e = (int64_t)r * slice + 4 * i;
                    if (live[u] && j % W == W - 1 && e + 4 <= t.n_params) {
                        pm[u] = *reinterpret_cast<const f32x4*>(t.p + e);
                        mm[u] = *reinterpret_cast<const f32x4*>(t.m + e);
                        vm[u] = *reinterpret_cast<const f32x4*>(t.v + e);
                    }
                }
            }
            ok = ll_poll<GB>(src, live, part, tag, p, dead, timeout_ticks);
            if (!ok) break;
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                if (!live[u]) continue;
                const int x = (j0 + u) % W;
                acc = x == 0 ? part[u] : acc + part[u];
                if (x == W - 1) {
                    const int64_t i = first + ((j0 + u) / W) * stride;
                    for (int y = 0; y + 1 < W; ++y) {
                        const int q = (r + 1 + y) % W;
                        ll_send(p.base[q] + ll_out_off(p, r, i), acc, tag);
                    }
                    consume((int64_t)r * slice + 4 * i, acc, pm[u], mm[u], vm[u]);
                }
            }
        }
    }

#ifdef TNN_AR_TRACE
    if (skip.trace && threadIdx.x == 0) skip.trace[2] = wall_clock64();
#endif
    // (C) the other owners' results -> caller's buffer (+ the optimizer update when ADAM).  A thread that lost a poll (timeout,
    // dead transport) updates nothing; with a peer missing that is every thread of every rank (see above).
    for (int j0 = 0; ok && W > 1 && j0 < items; j0 += UN) {
        f32x4 g[UN] = {};
        const char* gsrc[UN];
        bool glive[UN];
        int64_t at[UN];                               // element offset in buf (n = nothing to do)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u;
            const int q = j < items ? j % W : 0;
            glive[u] = j < items && q != r;           // this rank's own slice was consumed in stage (B)
            const int64_t i = first + (j < items ? j / W : 0) * stride;
            gsrc[u] = p.base[r] + ll_out_off(p, q, i);
            at[u] = glive[u] ? (int64_t)q * slice + 4 * i : n;
        }
        f32x4 pm[UN] = {}, mm[UN] = {}, vm[UN] = {};
        if constexpr (ADAM) {
            // the parameter / moment loads do not depend on the peers: issue them under the same wait
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (at[u] + 4 <= t.n_params) {
                    pm[u] = *reinterpret_cast<const f32x4*>(t.p + at[u]);
                    mm[u] = *reinterpret_cast<const f32x4*>(t.m + at[u]);
                    vm[u] = *reinterpret_cast<const f32x4*>(t.v + at[u]);
                }
            }
        }
        ok = ll_poll<UN>(gsrc, glive, g, tag, p, dead, timeout_ticks);
        if (!ok) break;
#pragma unroll
        for (int u = 0; u < UN; ++u) consume(at[u], g[u], pm[u], mm[u], vm[u]);
    }
#ifdef TNN_AR_TRACE
    if (skip.trace && threadIdx.x == 0) skip.trace[3] = wall_clock64();
#endif
    // A kernel whose OTHER workgroups produce the skipped ranges: gate_count of them draw their tag from THIS workgroup's
    // launch count and arrive at `gate` (an agent-scope counter in ordinary memory) once they have — the caller may advance
    // the count only after that.  Checked here, at the end, where it has long happened (one load off the critical path;
    // waiting for it BEFORE the first poll cost 1-2 us of atomic round trips).  Bounded like every other wait.  Block-uniform.
    if (gate != nullptr) {
        if (threadIdx.x == 0 && ok) {
            uint64_t t0 = 0;
            uint32_t polls = 0;
            while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gate_count) {
                __builtin_amdgcn_s_sleep(2);
                if ((++polls & 255u) == 0) {
                    const uint64_t now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if ((int64_t)(now - t0) > timeout_ticks) {
                        mark_dead(p, dead, 3, gate_count, __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), blockIdx.x, 0);
                        break;
                    }
                }
            }
        }
        __syncthreads();
    }

}

// ---- deferred statistics exchange (XchgCtx above): device side ------------------------------------------------------------
__device__ __forceinline__ size_t xchg_off(const int row, const int par, const int src) {
    return offsetof(Header, flag) + (size_t)row * FLAG_ROW + (size_t)par * MAXW * 16 + (size_t)src * 16;
}
// {M, S}: this shard's pair, the same in every thread of the workgroup on entry; the batch's pair (all ranks merged in rank
// order) in every thread on return.  `sender`: block-uniform, true in exactly ONE workgroup of the launch.  lds2: two floats of
// shared memory.  One workgroup barrier inside when world > 1; at world 1 nothing at all happens.  NT = threads per workgroup.
// world: the group's size as a kernel ARGUMENT (HeadMArgs::xw) — at world 1 the call must not cost a dependent load through xc.
template <int NT>
__device__ __forceinline__ void xchg_merge(const XchgCtx* __restrict__ xc, const int world, float& M, float& S, const bool sender,
                                           float* lds2) {
    if (world <= 1 || xc == nullptr) return;
    const int W = xc->peers.world;
    if (W <= 1) return;
    const int rank = xc->peers.rank, tid = threadIdx.x;
    const uint32_t tag = *xc->seq;
    const int par = (int)(tag & 1u);
    if (sender) {
        for (int s = tid; s < XCHG_ROWS * W; s += NT) {
            const int q = s % W, row = s / W;
            if (q != rank)
                ll_store16(xc->peers.base[q] + xchg_off(row, par, rank), u32x4{__float_as_uint(M), tag, __float_as_uint(S), tag});
        }
    }
    if (tid < 64) {
        float mq = -INFINITY, sq = 0.f;
        if (tid < W) {
            if (tid == rank) {
                mq = M; sq = S;
            } else if (__hip_atomic_load(xc->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                const char* src = xc->peers.base[rank] + xchg_off((int)(blockIdx.x % XCHG_ROWS), par, tid);
                uint64_t t0 = 0;
                uint32_t polls = 0;
                for (;;) {
                    u32x4 v;
                    ll_load16(v, src);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("" : "+v"(v));
                    if (v[1] == tag && v[3] == tag) { mq = __uint_as_float(v[0]); sq = __uint_as_float(v[2]); break; }
                    __builtin_amdgcn_s_sleep(1);
                    if ((++polls & 63u) == 0) {
                        const uint64_t now = wall_clock64();
                        if (t0 == 0) t0 = now;
                        if (__hip_atomic_load(xc->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        if ((int64_t)(now - t0) > xc->timeout_ticks) {
                            mark_dead(xc->peers, xc->dead, 2, tag, v[1], (uint32_t)tid, (uint32_t)par);
                            break;
                        }
                    }
                }
            }
        }
        // (a rank whose wait gave up contributes {-inf, 0}: the update behind this launch is discarded with the dead word)
        const float Mg = tnn::wave_max_dpp(mq);
        const float Sg = tnn::wave_sum_dpp(mq > -INFINITY ? sq * expf(mq - Mg) : 0.f);
        if (tid == 0) { lds2[0] = Mg; lds2[1] = Sg; }
    }
    __syncthreads();
    M = lds2[0];
    S = lds2[1];
}

}  // namespace p2p

const p2p::XchgCtx* p2p_xchg_ctx();                          // device pointer, NULL when the transport is not enabled
int p2p_ranks_on_my_device();                                // ranks of the group whose region lives on this rank's GPU (>= 1); 0 = no group
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
// bulk collectives over the mapped regions (tnn_p2p.hip; staging reserved with tnn_p2p_set_bulk_bytes before the group was made)
bool p2p_can_bulk(int64_t n_per_rank, int dtype, bool sum);  // enabled, staging exists, whole 16-byte units, (sum: bf16 / f32)
int p2p_reduce_scatter(const void* send, void* recv, int64_t n_per_rank, int dtype);
int p2p_allgather_bulk(const void* send, void* recv, int64_t n_per_rank, int dtype);
}  // namespace tnn
