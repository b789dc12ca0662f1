// C1/C2 over xGMI peer-to-peer stores: a low-latency transport under tnn_allreduce / tnn_allgather for the
// messages the MNIST-size data-parallel step exchanges (a 0.94 MB gradient arena and one {max, sum-exp} pair
// per rank; examples/mnist/run.py:82-83 is where the exchange sits, core/losses.py:26-27 is why the second one
// exists).  At 32 us per training step the two RCCL calls ARE the multi-GPU cost, so this path is latency-first:
//
//   * every rank owns one UNCACHED device region (hipDeviceMallocUncached: remote stores and local reads bypass
//     L2, so data written by a peer inside a running kernel is visible without a kernel boundary), exported
//     once with hipIpcGetMemHandle and mapped by every peer;
//   * all cross-GPU traffic is PUSHED (posted xGMI writes; nothing waits for a remote read round trip);
//   * all-reduce = ONE kernel, two-stage: (A) rank r stores its copy of slice p into peer p's recv slots [r];
//     (B) rank r sums the W copies of its own slice IN RANK ORDER and stores the result slice into every peer's out
//     slots; (C) out slots -> the caller's buffer.  Each slice is reduced by exactly one rank and broadcast, so all
//     ranks end up with bit-identical sums (replicas cannot drift apart); per link and direction: 2 x n/W elements;
//   * no barriers between the stages (round 3): every payload word travels next to a tag in one 8-byte half of a
//     16-byte store and the receiver polls the data itself (ll_send / ll_poll below) — one fabric latency per stage
//     instead of three dependent ones (store-ack wait, flag store -> poll, data load).  Tags grow monotonically from a
//     per-block launch count kept in DEVICE memory, so the kernel is replayable from a hipGraph with fixed arguments
//     and never needs a reset; the small all-gather still uses a flag exchange (exchange_flags, tnn_p2p.h);
//   * (C) can carry the optimizer: the reduced gradient is in registers there, so Adam is applied on the spot;
//   * spins are bounded by the constant 100 MHz clock: on timeout the kernel sets a sticky `dead` word (device) and
//     its host-pinned mirror, and from then on NOTHING is consumed: a workgroup whose barrier failed — in this or any
//     later launch, captured graph replays included — skips the reduction, the copy-out and the optimizer tail, so
//     the caller's buffer, the parameters and the Adam moments keep their pre-collective contents instead of being
//     updated from partial sums; the next host-side call into the transport returns an error (p2p_refuse_if_failed
//     reads the mirror without a stream sync) and tnn_p2p_status() reports it — a lost peer is a loud error, neither
//     a hung GPU nor silently diverging replicas.
//
// Buffer reuse is safe without a handshake: block b of a rank finishes call k only when it has received EVERY out slot of
// its sub-range, which each peer sends only after reading all its recv slots of that sub-range — so call k+1 may
// overwrite the recv slots; out slots of call k+1 are written after the writer received the stage-A data of call k+1 from
// every rank, which a rank sends only after its kernel k (stage C included) has finished in stream order.
// The small all-gather double-buffers its slots on epoch parity for the same reason.
#include <string.h>

#include "tnn_internal.h"
#include "tnn_p2p.h"

namespace {

using namespace tnn::p2p;

// the local (cached) block: [MAXB + 1] epochs | 64 spare bytes (the sticky dead word at + 28, the deferred statistics
// exchange's launch sequence at + 32) | the gate counters
constexpr size_t LOCAL_GATES = ((MAXB + 1) * sizeof(uint32_t) + 64 + 255) / 256 * 256;
constexpr int BULK_BLOCKS = 64;                  // workgroups of every bulk-collective launch (flag rows Header::flag[1][b])
constexpr size_t LOCAL_BULK = LOCAL_GATES + (size_t)MAXB * GATE_STRIDE * sizeof(uint32_t);    // [BULK_BLOCKS] launch counts
constexpr size_t LOCAL_BYTES = LOCAL_BULK + (size_t)BULK_BLOCKS * sizeof(uint32_t);

struct State {
    bool open = false, enabled = false;
    Peers p = {};
    char* own = nullptr;
    void* mapped[MAXW] = {};                      // hipIpcOpenMemHandle results (NULL for self)
    uint32_t* epoch = nullptr;                    // [MAXB + 1] per-block epochs, the all-gather / statistics-exchange epoch
                                                  // (local, cached)
    int* dead = nullptr;                          // sticky timeout word (local)
    uint32_t* xchg_seq = nullptr;                 // launch sequence of the deferred statistics exchange (local; tnn_p2p.h: XchgCtx)
    XchgCtx* xchg_dev = nullptr;                  // device copy of the XchgCtx the head kernels take a pointer to
    int* host_dead = nullptr;                     // its host-pinned mirror (hipHostMalloc, mapped): host address
    int64_t max_floats = 0;
    int64_t timeout_ticks = 0;
    int grid = MAXB;                              // workgroups of EVERY all-reduce launch (see p2p_allreduce_kernel)
    size_t bulk_off = 0, bulk_slot = 0;           // bulk staging: region + bulk_off + (parity * W + source) * bulk_slot
    uint32_t* bulk_epoch = nullptr;               // [BULK_BLOCKS] launch counts of the bulk collectives' workgroups (local)
    int32_t devid[4] = {};                        // this GPU's identity (Header::devid)
    int on_my_device = 1;                         // ranks whose region lives on this GPU, this one included
} S;

int device_identity(int32_t (&id)[4]) {
    int dev = 0, dom = 0, bus = 0, devn = 0;
    TNN_CHECK_HIP(hipGetDevice(&dev));
    TNN_CHECK_HIP(hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, dev));
    TNN_CHECK_HIP(hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, dev));
    TNN_CHECK_HIP(hipDeviceGetAttribute(&devn, hipDeviceAttributePciDeviceId, dev));
    id[0] = dom; id[1] = bus; id[2] = devn; id[3] = 1;
    return 0;
}

// buf[0:n] <- sum over ranks, in place.  slice = floats per rank slice (multiple of 4, W*slice >= n).
//   (A) my copy of slice q -> rank q's recv slots [me]         (tagged stores, nothing waits)
//   (B) poll my recv slots of all W sources, sum IN RANK ORDER, tagged stores of the result into every rank's out slots
//   (C) poll my out slots of all W slices -> caller's buffer, with Adam applied to the registers when ADAM
// Slot -> workgroup map: float4 element i of a slice belongs to workgroup (i / 512) % G, thread i % 512, whatever the
// message size, and the grid is ALWAYS G workgroups — so a slot is only ever written and read on behalf of one workgroup
// index, and that workgroup's launch count (device memory, advanced by every launch, hipGraph-replayable) is a strictly
// increasing tag for it: a stale slot can never carry the tag a later call polls for.  A rank's launch k + 1 starts only
// after its launch k has received EVERY out slot it owns a thread for, which each peer sends only after it has read all its
// recv slots of that workgroup — so launch k + 1 may overwrite both halves without any further handshake.  A missing peer
// starves stage B on every rank (each sum needs all W copies), nothing reaches any out slot, and no rank updates anything.
template <bool ADAM>
__global__ __launch_bounds__(THREADS) void p2p_allreduce_kernel(Peers p, float* __restrict__ buf, int64_t n,
                                                                int64_t slice, uint32_t* __restrict__ epoch,
                                                                int* dead, int64_t timeout_ticks, AdamTail t) {
    const int b = blockIdx.x, G = gridDim.x;
    const uint32_t tag = epoch[b] + 1;
    // this thread's elements of a slice: i_k = (b + k G) 512 + thread
    allreduce_body<ADAM>(p, buf, n, slice, tag, dead, timeout_ticks, t, (int64_t)b * THREADS + threadIdx.x, (int64_t)G * THREADS,
                         SkipRanges{});
    __syncthreads();                                      // every thread has read epoch[b]
    if (threadIdx.x == 0) epoch[b] = tag;
}

__global__ __launch_bounds__(MAXB) void p2p_level_tags_kernel(uint32_t* __restrict__ epoch) {
    __shared__ uint32_t top;
    if (threadIdx.x == 0) top = 0;
    __syncthreads();
    atomicMax(&top, epoch[threadIdx.x]);
    __syncthreads();
    epoch[threadIdx.x] = top;
}

// recv[q][0:words] <- rank q's send[0:words]  (words <= 64), one workgroup
__global__ __launch_bounds__(THREADS) void p2p_allgather_kernel(Peers p, const uint32_t* __restrict__ send,
                                                                uint32_t* __restrict__ recv, int words,
                                                                uint32_t* __restrict__ epoch, int* dead,
                                                                int64_t timeout_ticks) {
    const int W = p.world, r = p.rank;
    const uint32_t e = *epoch;
    const int par = e & 1;
    const size_t slots = offsetof(Header, ag_slot) + (size_t)par * MAXW * AG_BYTES;
    for (int t = threadIdx.x; t < W * words; t += THREADS) {
        const int q = t / words, w = t % words;
        store_sys(reinterpret_cast<uint32_t*>(p.base[q] + slots + (size_t)r * AG_BYTES) + w, send[w]);
    }
    const bool ok = exchange_flags(p, offsetof(Header, ag_flag), e + 1, dead, timeout_ticks);
    for (int t = threadIdx.x; ok && t < W * words; t += THREADS) {
        const int q = t / words, w = t % words;
        uint32_t v[1] = {0u};
        load_sys(v[0], reinterpret_cast<const uint32_t*>(p.base[r] + slots + (size_t)q * AG_BYTES) + w);
        loads_landed(v);
        recv[t] = v[0];
    }
    if (threadIdx.x == 0) *epoch = e + 1;
}

// ---- bulk collectives: reduce-scatter (bf16 / f32 sums) and all-gather of bandwidth-sized messages over the same mapped
// regions, for groups WITHOUT an RCCL communicator (TNN_COMM=xgmi; RCCL refuses ranks that share a device, and the
// sharded-optimizer step of configs[4] — csrc/tnn_mlp.cpp mlp16_step_zero: reduce-scatter of the bf16 weight gradient, Adam
// on the owned rows, all-gather of the bf16 rows — must be runnable with rank > 0 on a one-GPU box).  Direct exchange, every
// byte crosses one link once: (A) rank r pushes its copy of shard q into rank q's staging slot [r] (16-byte write-through
// stores); one flag barrier per workgroup (exchange_flags on Header::flag[1][b]: workgroup b of every rank owns the same 16-byte
// units, so it only needs workgroup b of the peers); (B) rank q sums the W copies of its shard in rank order — fp32
// accumulation, ONE rounding to the wire type at the end — into recv.  The all-gather pushes the rank's shard into every
// peer's slot [r] and copies the W slots out.  Slots are double-buffered on the launch count's parity: passing barrier k + 1
// means every peer has finished launch k, so launch k + 2 may overwrite what launch k read.  Messages larger than a slot
// go in chunks (one launch each).  hipGraph-replayable: launch counts live in device memory.
template <int KIND>          // 0: reduce-scatter bf16, 1: reduce-scatter f32, 2: all-gather (bytes)
__global__ __launch_bounds__(THREADS) void p2p_bulk_kernel(Peers p, const char* __restrict__ send, char* __restrict__ recv,
                                                           int64_t shard_bytes, int64_t off_bytes, int64_t chunk_bytes,
                                                           size_t bulk_off, size_t bulk_slot, uint32_t* __restrict__ epoch,
                                                           int* dead, int64_t timeout_ticks) {
    const int W = p.world, r = p.rank, b = blockIdx.x;
    const uint32_t e = epoch[b];
    const size_t par_off = bulk_off + (size_t)(e & 1u) * W * bulk_slot;
    const int64_t units = chunk_bytes / 16, first = (int64_t)b * THREADS + threadIdx.x, stride = (int64_t)gridDim.x * THREADS;
    // (A) push
    for (int64_t u = first; u < units; u += stride) {
        if constexpr (KIND == 2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(send + off_bytes + u * 16);
            for (int y = 0; y < W; ++y) {
                const int q = (r + 1 + y) % W;
                store_sys(reinterpret_cast<float*>(p.base[q] + par_off + (size_t)r * bulk_slot + u * 16), v);
            }
        } else {
            for (int y = 0; y < W; ++y) {
                const int q = (r + 1 + y) % W;
                const f32x4 v = *reinterpret_cast<const f32x4*>(send + (int64_t)q * shard_bytes + off_bytes + u * 16);
                store_sys(reinterpret_cast<float*>(p.base[q] + par_off + (size_t)r * bulk_slot + u * 16), v);
            }
        }
    }
    const bool ok = exchange_flags(p, offsetof(Header, flag) + ((size_t)MAXB + b) * FLAG_ROW, e + 1, dead, timeout_ticks);
    // (B) reduce / copy out
    const __amdgpu_buffer_rsrc_t stage = __builtin_amdgcn_make_buffer_rsrc(p.base[r] + par_off, 0, (uint32_t)((size_t)W * bulk_slot), 0x00020000);
    for (int64_t u = first; ok && u < units; u += stride) {
        // cache-bypassing loads the COMPILER can see (buffer loads with the sc0 sc1 policy bits, as in tnn_head_stats.h): behind a
        // conditional inline-asm load the compiler inserts a register copy that reads the destination before the data has landed
        // (measured: wrong sums).  Loops unrolled to MAXW with a predicate: the array stays in registers.
        f32x4 v[MAXW];
#pragma unroll
        for (int q = 0; q < MAXW; ++q)
            if (q < W)
                v[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(stage, (uint32_t)(u * 16), (uint32_t)((size_t)q * bulk_slot), 17));
        if constexpr (KIND == 2) {
#pragma unroll
            for (int q = 0; q < MAXW; ++q)
                if (q < W) *reinterpret_cast<f32x4*>(recv + (int64_t)q * shard_bytes + off_bytes + u * 16) = v[q];
        } else if constexpr (KIND == 1) {
            f32x4 acc = v[0];
#pragma unroll
            for (int q = 1; q < MAXW; ++q)
                if (q < W) acc += v[q];
            *reinterpret_cast<f32x4*>(recv + off_bytes + u * 16) = acc;
        } else {
            float acc[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t w = __float_as_uint(v[0][k]);
                acc[2 * k] = __uint_as_float(w << 16);
                acc[2 * k + 1] = __uint_as_float(w & 0xffff0000u);
            }
#pragma unroll
            for (int q = 1; q < MAXW; ++q) {
                if (q >= W) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t w = __float_as_uint(v[q][k]);
                    acc[2 * k] += __uint_as_float(w << 16);
                    acc[2 * k + 1] += __uint_as_float(w & 0xffff0000u);
                }
            }
            f32x4 out;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t lo = __float_as_uint(acc[2 * k]), hi = __float_as_uint(acc[2 * k + 1]);
                lo = (lo + 0x7fffu + ((lo >> 16) & 1u)) >> 16;                     // round to nearest even (finite sums)
                hi = (hi + 0x7fffu + ((hi >> 16) & 1u)) & 0xffff0000u;
                out[k] = __uint_as_float(hi | lo);
            }
            *reinterpret_cast<f32x4*>(recv + off_bytes + u * 16) = out;
        }
    }
    __syncthreads();                                      // every thread has read epoch[b]
    if (threadIdx.x == 0) epoch[b] = e + 1;
}

// Start-up check of the deferred statistics exchange (xchg_merge): every workgroup of a 64-workgroup launch merges the ranks' pairs
// exactly as the head kernels do and writes what it got; a one-thread launch in front advances the sequence number as the forward
// launch of a step does.
__global__ void p2p_xchg_bump_kernel(uint32_t* seq) { *seq += 1u; }
__global__ __launch_bounds__(256) void p2p_xchg_test_kernel(const XchgCtx* xc, int world, float M, float S, float* __restrict__ out) {
    __shared__ float xm[2];
    xchg_merge<256>(xc, world, M, S, blockIdx.x == 0, xm);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = M; out[2 * blockIdx.x + 1] = S; }
}

int esize(int dtype) {
    switch (dtype) {
        case TNN_F32: return 4;
        case TNN_F64: case TNN_I64: return 8;
        case TNN_U8: return 1;
    }
    return 0;
}

}  // namespace

namespace tnn {

bool p2p_world(int* rank, int* world) {
    if (!S.open) return false;
    if (rank) *rank = S.p.rank;
    if (world) *world = S.p.world;
    return true;
}

bool p2p_failed() { return S.open && S.host_dead && *(volatile int*)S.host_dead != 0; }

int p2p_refuse_if_failed(const char* who) {
    if (!p2p_failed()) return 0;
    tnn::set_error("%s: an xGMI peer-to-peer barrier timed out earlier (a peer was missing for longer than "
                   "TNN_P2P_TIMEOUT_MS); that collective and every one since was discarded and left buffers and "
                   "parameters untouched.  Disable the transport on EVERY rank (tnn_p2p_enable(0)) to continue on "
                   "RCCL, or restart the job", who);
    return 3;
}

bool p2p_can_allreduce(int64_t n, int dtype, int rop) {
    return S.enabled && dtype == TNN_F32 && rop == TNN_RSUM && n > 0 && n <= S.max_floats;
}

static int launch_allreduce(float* buf, int64_t n, const AdamTail* tail) {
    if (int rc = p2p_refuse_if_failed("tnn_allreduce")) return rc;
    const int W = S.p.world;
    int64_t slice = (n + W - 1) / W;
    slice = (slice + 3) / 4 * 4;
    // ALWAYS the same grid (the slot -> workgroup map and the per-workgroup tags depend on it): S.grid, fixed when the group
    // is created (TNN_P2P_BLOCKS, default 128: one workgroup per 512 float4 of the 0.94 MB arena) and changed only through
    // tnn_p2p_tune, which re-levels the tags.  Workgroups without an element of a small message just advance their tag.
    const int64_t blocks = S.grid;
    if (tail)
        hipLaunchKernelGGL(p2p_allreduce_kernel<true>, dim3((unsigned)blocks), dim3(THREADS), 0, tnn::stream(), S.p, buf,
                           n, slice, S.epoch, S.dead, S.timeout_ticks, *tail);
    else
        hipLaunchKernelGGL(p2p_allreduce_kernel<false>, dim3((unsigned)blocks), dim3(THREADS), 0, tnn::stream(), S.p, buf,
                           n, slice, S.epoch, S.dead, S.timeout_ticks, AdamTail{});
    TNN_LAUNCH_OK();
    return 0;
}

int p2p_allreduce(float* buf, int64_t n) { return launch_allreduce(buf, n, nullptr); }

int p2p_allreduce_adam(float* buf, int64_t n, float* p, float* m, float* v, int64_t n_params, double lr, double b1,
                       double b2, double eps, const double* pows, int64_t scalar_index, float* scalar_dst) {
    AdamTail t;
    t.p = p; t.m = m; t.v = v; t.n_params = n_params;
    t.lr = (float)lr; t.b1 = (float)b1; t.b2 = (float)b2; t.eps = (float)eps;
    t.pows = pows;
    t.scalar_index = scalar_dst ? scalar_index : -1;
    t.scalar_dst = scalar_dst;
    return launch_allreduce(buf, n, &t);
}

bool p2p_launch_ctx(p2p::LaunchCtx* ctx) {
    if (!S.enabled || p2p_failed()) return false;
    ctx->peers = S.p;
    ctx->ag_epoch = S.epoch + MAXB;
    ctx->dead = S.dead;
    ctx->timeout_ticks = S.timeout_ticks;
    ctx->ar_epoch = S.epoch;
    ctx->ar_gate = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(S.epoch) + LOCAL_GATES);
    ctx->ar_grid = S.grid;
    ctx->xchg_seq = S.xchg_seq;
    return true;
}

const p2p::XchgCtx* p2p_xchg_ctx() { return S.enabled && !p2p_failed() ? S.xchg_dev : nullptr; }
int p2p_ranks_on_my_device() { return S.open ? S.on_my_device : 0; }

bool p2p_can_allgather(int64_t n_per_rank, int dtype) {
    const int64_t bytes = n_per_rank * esize(dtype);
    return S.enabled && bytes > 0 && bytes <= AG_BYTES && bytes % 4 == 0;
}

int p2p_allgather(const void* send, void* recv, int64_t n_per_rank, int dtype) {
    if (int rc = p2p_refuse_if_failed("tnn_allgather")) return rc;
    const int words = (int)(n_per_rank * esize(dtype) / 4);
    hipLaunchKernelGGL(p2p_allgather_kernel, dim3(1), dim3(THREADS), 0, tnn::stream(), S.p, (const uint32_t*)send,
                       (uint32_t*)recv, words, S.epoch + MAXB, S.dead, S.timeout_ticks);
    TNN_LAUNCH_OK();
    return 0;
}

bool p2p_can_bulk(int64_t n_per_rank, int dtype, bool sum) {
    if (!S.enabled || S.bulk_slot == 0 || n_per_rank <= 0) return false;
    const int64_t es = dtype == TNN_BF16 ? 2 : esize(dtype);
    if (es == 0 || (n_per_rank * es) % 16 != 0) return false;
    return !sum || dtype == TNN_BF16 || dtype == TNN_F32;
}

static int bulk_launch(int kind, const void* send, void* recv, int64_t shard_bytes) {
    if (int rc = p2p_refuse_if_failed(kind == 2 ? "tnn_allgather" : "tnn_reduce_scatter")) return rc;
    TNN_REQUIRE(((reinterpret_cast<uintptr_t>(send) | reinterpret_cast<uintptr_t>(recv)) & 15) == 0,
                "peer-to-peer bulk collective: buffers must be 16-byte aligned");
    const int64_t slot = (int64_t)S.bulk_slot;
    for (int64_t off = 0; off < shard_bytes; off += slot) {
        const int64_t chunk = shard_bytes - off < slot ? shard_bytes - off : slot;
#define TNN_BULK(K)                                                                                                          \
        hipLaunchKernelGGL(p2p_bulk_kernel<K>, dim3(BULK_BLOCKS), dim3(THREADS), 0, tnn::stream(), S.p, (const char*)send,    \
                           (char*)recv, shard_bytes, off, chunk, S.bulk_off, S.bulk_slot, S.bulk_epoch, S.dead, S.timeout_ticks)
        if (kind == 0) TNN_BULK(0);
        else if (kind == 1) TNN_BULK(1);
        else TNN_BULK(2);
#undef TNN_BULK
        TNN_LAUNCH_OK();
    }
    return 0;
}

int p2p_reduce_scatter(const void* send, void* recv, int64_t n_per_rank, int dtype) {
    return bulk_launch(dtype == TNN_BF16 ? 0 : 1, send, recv, n_per_rank * (dtype == TNN_BF16 ? 2 : 4));
}

int p2p_allgather_bulk(const void* send, void* recv, int64_t n_per_rank, int dtype) {
    return bulk_launch(2, send, recv, n_per_rank * (dtype == TNN_BF16 ? 2 : esize(dtype)));
}

}  // namespace tnn

static int64_t g_bulk_request = 0;            // tnn_p2p_set_bulk_bytes: staging bytes per (parity, source) slot of the NEXT group

extern "C" {

int tnn_p2p_set_bulk_bytes(int64_t slot_bytes) {
    TNN_REQUIRE(slot_bytes >= 0 && slot_bytes % 4096 == 0 && slot_bytes <= ((int64_t)1 << 27),
                "tnn_p2p_set_bulk_bytes: a multiple of 4096 up to 128 MiB (0 = no bulk staging)");
    g_bulk_request = slot_bytes;
    return 0;
}


int tnn_p2p_create(int rank, int world, int64_t max_bytes, void* handle64) {
    TNN_NEED_INIT();
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
    TNN_REQUIRE(!S.open, "tnn_p2p_create: already created");
    TNN_REQUIRE(world >= 1 && world <= MAXW && rank >= 0 && rank < world, "tnn_p2p_create: rank %d / world %d (max %d)",
                rank, world, MAXW);
    TNN_REQUIRE(max_bytes >= 16 && handle64, "tnn_p2p_create: bad arguments");
    const int64_t max_floats = max_bytes / 4;
    int64_t cap = (max_floats + world - 1) / world;
    cap = (cap + 3) / 4 * 4;
    const size_t ll_bytes = HEADER_BYTES + (size_t)2 * world * cap * 8;   // recv + out halves, every payload word next to its tag
    const size_t bulk_off = (ll_bytes + 4095) / 4096 * 4096;
    const size_t bytes = bulk_off + (size_t)2 * world * (size_t)g_bulk_request;      // + the bulk collectives' staging slots
    void* region = nullptr;
    TNN_CHECK_HIP(hipExtMallocWithFlags(&region, bytes, hipDeviceMallocUncached));
    TNN_CHECK_HIP(hipMemset(region, 0, bytes));
    int32_t devid[4];
    if (int rc = device_identity(devid)) return rc;
    TNN_CHECK_HIP(hipMemcpy((char*)region + offsetof(Header, devid), devid, sizeof(devid), hipMemcpyHostToDevice));
    void* local = nullptr;
    TNN_CHECK_HIP(hipMalloc(&local, LOCAL_BYTES));
    TNN_CHECK_HIP(hipMemset(local, 0, LOCAL_BYTES));
    void* host_dead = nullptr;
    void* host_dead_dev = nullptr;
    TNN_CHECK_HIP(hipHostMalloc(&host_dead, 64, hipHostMallocMapped));
    memset(host_dead, 0, 64);
    TNN_CHECK_HIP(hipHostGetDevicePointer(&host_dead_dev, host_dead, 0));
    TNN_CHECK_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    TNN_CHECK_HIP(hipIpcGetMemHandle(&h, region));
    memcpy(handle64, &h, sizeof(h));
    S = State();
    S.own = (char*)region;
    S.host_dead = (int*)host_dead;
    S.p.dead_host = (int*)host_dead_dev;
    S.epoch = (uint32_t*)local;
    S.dead = (int*)((char*)local + (MAXB + 1) * sizeof(uint32_t) + 28);
    S.xchg_seq = (uint32_t*)((char*)local + (MAXB + 1) * sizeof(uint32_t) + 32);
    S.p.rank = rank;
    S.p.world = world;
    S.p.slice_cap = cap;
    S.p.poll_gap = getenv("TNN_P2P_POLL_GAP") ? atoi(getenv("TNN_P2P_POLL_GAP")) : 1;
    // (measured at world 1, tools/probes/dp_poll_ab.py: 32 units before the first stage-B poll of the fused launch: - 0.3 us per
    // step at 128 rows; the gap between polls makes no difference between 1 and 16 units)
    S.p.poll_first = getenv("TNN_P2P_POLL_FIRST") ? atoi(getenv("TNN_P2P_POLL_FIRST")) : 32;
    S.max_floats = max_floats;
    memcpy(S.devid, devid, sizeof(devid));
    S.bulk_off = bulk_off;
    S.bulk_slot = (size_t)g_bulk_request;
    S.bulk_epoch = (uint32_t*)((char*)local + LOCAL_BULK);
    const char* to = getenv("TNN_P2P_TIMEOUT_MS");
    const double ms = to ? atof(to) : 20000.0;
    S.timeout_ticks = (int64_t)(ms * 1e5);                      // wall_clock64(): 100 MHz
    const char* nb = getenv("TNN_P2P_BLOCKS");
    S.grid = nb && atoi(nb) >= 1 && atoi(nb) <= MAXB ? atoi(nb) : MAXB;
    S.open = true;
    return 0;
}

int tnn_p2p_connect(const void* handles) {
    TNN_NEED_INIT();
    TNN_REQUIRE(S.open && !S.enabled, "tnn_p2p_connect: create first (and connect once)");
    TNN_REQUIRE(handles != nullptr, "tnn_p2p_connect: handles is NULL");
    for (int q = 0; q < S.p.world; ++q) {
        if (q == S.p.rank) {
            S.p.base[q] = S.own;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)q * 64, sizeof(h));
        void* ptr = nullptr;
        TNN_CHECK_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        S.mapped[q] = ptr;
        S.p.base[q] = (char*)ptr;
    }
    // which peers live on THIS GPU (the tests' shared-GPU groups; never in a one-process-per-GPU job): kernels in which EVERY
    // workgroup waits for a peer are only launched when all such ranks' launches fit the device together (tnn_mlp_head_bwd_xchg_fits)
    S.on_my_device = 1;
    for (int q = 0; q < S.p.world; ++q) {
        if (q == S.p.rank) continue;
        int32_t theirs[4] = {};
        TNN_CHECK_HIP(hipMemcpy(theirs, S.p.base[q] + offsetof(Header, devid), sizeof(theirs), hipMemcpyDeviceToHost));
        if (theirs[3] == 1 && memcmp(theirs, S.devid, sizeof(theirs)) == 0) ++S.on_my_device;
    }
    // what the head kernels of the deferred statistics exchange read through a pointer (fixed from here on)
    XchgCtx xc;
    xc.peers = S.p;
    xc.seq = S.xchg_seq;
    xc.dead = S.dead;
    xc.timeout_ticks = S.timeout_ticks;
    TNN_CHECK_HIP(hipMalloc((void**)&S.xchg_dev, sizeof(XchgCtx)));
    TNN_CHECK_HIP(hipMemcpy(S.xchg_dev, &xc, sizeof(XchgCtx), hipMemcpyHostToDevice));
    S.enabled = true;
    return 0;
}

int tnn_p2p_enable(int on) {
    TNN_REQUIRE(S.open && S.p.base[S.p.rank] != nullptr, "tnn_p2p_enable: not connected");
    S.enabled = on != 0;
    return 0;
}

int tnn_p2p_tune(int allreduce_blocks) {
    TNN_REQUIRE(S.open, "tnn_p2p_tune: no peer group");
    TNN_REQUIRE(allreduce_blocks >= 0 && allreduce_blocks <= MAXB, "tnn_p2p_tune: blocks must be in [0, %d]", MAXB);
    // every rank must use the same value (workgroup b of one rank feeds workgroup b of the others).  A different grid maps
    // slots to different workgroups: level every workgroup's tag at the maximum first, so no stale slot can match
    const int grid = allreduce_blocks == 0 ? MAXB : allreduce_blocks;
    if (grid != S.grid) {
        hipLaunchKernelGGL(p2p_level_tags_kernel, 1, MAXB, 0, tnn::stream(), S.epoch);
        TNN_LAUNCH_OK();
        S.grid = grid;
    }
    return 0;
}

int tnn_p2p_status(int* connected, int* enabled, int* dead) {
    if (connected) *connected = S.open && S.p.base[S.p.rank] != nullptr;
    if (enabled) *enabled = S.enabled;
    if (dead) {
        *dead = 0;
        if (S.open) {
            TNN_CHECK_HIP(hipStreamSynchronize(tnn::stream()));
            TNN_CHECK_HIP(hipMemcpy(dead, S.dead, sizeof(int), hipMemcpyDeviceToHost));
        }
    }
    return 0;
}

int tnn_p2p_xchg_selftest(double m_mine, double s_mine, void* out_pairs_f32) {
    // out_pairs_f32: device [64][2] floats — the pair every one of 64 workgroups ended up with (all equal to the merge of the ranks'
    // {m_mine, s_mine} when the exchange works).  Collective: every rank calls it once per round.
    TNN_NEED_INIT();
    TNN_REQUIRE(S.open && S.enabled && S.xchg_dev != nullptr && out_pairs_f32 != nullptr, "tnn_p2p_xchg_selftest: the transport is not enabled");
    if (int rc = tnn::p2p_refuse_if_failed("tnn_p2p_xchg_selftest")) return rc;
    hipLaunchKernelGGL(p2p_xchg_bump_kernel, 1, 1, 0, tnn::stream(), S.xchg_seq);
    hipLaunchKernelGGL(p2p_xchg_test_kernel, 64, 256, 0, tnn::stream(), S.xchg_dev, S.p.world, (float)m_mine, (float)s_mine, (float*)out_pairs_f32);
    TNN_LAUNCH_OK();
    return 0;
}

int tnn_p2p_debug(int* words16) {
    // the host mirror as it is: [0] 0 / which wait gave up, [1] value it expected, [2] last value it saw, [3] peer or
    // workgroup, [4] wait-specific detail (flag row / slot parity)
    TNN_REQUIRE(words16 != nullptr, "tnn_p2p_debug: words16 is NULL");
    for (int i = 0; i < 16; ++i) words16[i] = S.open && S.host_dead ? ((volatile int*)S.host_dead)[i] : 0;
    return 0;
}

int tnn_p2p_guard_updates(int on) {
    // While on, every optimizer-update kernel this library launches (SGD / Adam / the other four, fp32 and bf16 master)
    // first looks at the transport's sticky `dead` word and leaves parameters, state and beta powers untouched when it
    // is set: an update that would consume a discarded collective is discarded with it, also when the update is a
    // launch of its own behind the all-reduce (tnn_mlp_step_sharded brackets itself with this).
    tnn::set_update_guard(on && S.open ? S.dead : nullptr);
    return 0;
}

int tnn_p2p_poll_failed(int* failed) {
    // the host mirror of the sticky word: no stream synchronisation, safe to call before every graph replay
    if (failed) *failed = tnn::p2p_failed() ? 1 : 0;
    return 0;
}

int tnn_p2p_destroy(void) {
    if (!S.open) return 0;
    if (tnn::initialised()) (void)hipStreamSynchronize(tnn::stream());
    for (int q = 0; q < S.p.world; ++q)
        if (S.mapped[q]) (void)hipIpcCloseMemHandle(S.mapped[q]);
    (void)hipFree(S.own);
    (void)hipFree(S.epoch);
    if (S.xchg_dev) (void)hipFree(S.xchg_dev);
    if (S.host_dead) (void)hipHostFree(S.host_dead);
    S = State();
    return 0;
}

}  // extern "C"
