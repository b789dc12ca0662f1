// C1/C2: RCCL over xGMI.  New relative to the reference, which has no communication at all
// (SURVEY F1): one process per GPU, the flat gradient arena is all-reduced in place on the library
// stream between backward() and the optimizer update (the hook sits between examples/mnist/run.py:82
// and :83), and the whole-batch softmax (core/losses.py:26-27) exchanges one {max, sum-exp} pair per
// rank.  librccl is dlopen()ed on first use so that single-GPU users, the CPU container and the
// symbol test never need it, and so that a process which already loaded torch's copy reuses it.
// tnn_allreduce / tnn_allgather are the front for TWO transports: messages the xGMI peer-to-peer path of
// tnn_p2p.hip accepts (f32 sums up to its mapped capacity, all-gathers up to 256 B per rank) go there when it
// is enabled; everything else goes to RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <vector>

#include "tnn_internal.h"
#include "tnn_p2p.h"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                                  hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};
Rccl R;
hipStream_t g_comm_stream = nullptr;            // bucketed all-reduces run here (tnn_allreduce_async)
// "bucket done" events not yet joined, oldest first.  One data-parallel step issues and joins all of its own buckets
// inside ONE C call (tnn_mlp_step_sharded; its error paths drain through tnn_comm_join), so the list is empty between
// steps and trainers cannot pick up each other's events.  Events are recycled through g_event_pool: none is created or
// destroyed per bucket once the pool has warmed up.
std::vector<hipEvent_t>* g_pending = nullptr;
std::vector<hipEvent_t>* g_event_pool = nullptr;

int take_event(hipEvent_t* e) {
    if (g_event_pool && !g_event_pool->empty()) {
        *e = g_event_pool->back();
        g_event_pool->pop_back();
        return 0;
    }
    TNN_CHECK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return 0;
}
void give_event(hipEvent_t e) {
    // re-recording an event later does not disturb waits already enqueued on its previous recording
    if (!g_event_pool) g_event_pool = new std::vector<hipEvent_t>();
    g_event_pool->push_back(e);
}

int load_rccl() {
    if (R.so) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        R.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (R.so) break;
    }
    TNN_REQUIRE(R.so != nullptr, "tnn_comm: cannot dlopen librccl (%s)", dlerror());
#define SYM(field, name)                                                         \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.so, name));            \
    TNN_REQUIRE(R.field != nullptr, "tnn_comm: librccl lacks symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(ReduceScatter, "ncclReduceScatter");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return 0;
}

#define TNN_CHECK_NCCL(expr)                                                              \
    do {                                                                                  \
        ncclResult_t r__ = (expr);                                                        \
        if (r__ != ncclSuccess) {                                                         \
            tnn::set_error("%s -> %s", #expr, R.GetErrorString ? R.GetErrorString(r__) : "?"); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

int nccl_type(int dtype, ncclDataType_t* t) {
    switch (dtype) {
        case TNN_F32: *t = ncclFloat32; return 0;
        case TNN_F64: *t = ncclFloat64; return 0;
        case TNN_I64: *t = ncclInt64; return 0;
        case TNN_U8: *t = ncclUint8; return 0;
        case TNN_BF16: *t = ncclBfloat16; return 0;
    }
    tnn::set_error("tnn_comm: unknown dtype %d", dtype);
    return 2;
}

}  // namespace

extern "C" {

int tnn_comm_unique_id(void* id128) {
    TNN_NEED_INIT();
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    TNN_CHECK_NCCL(R.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int tnn_comm_init(int rank, int world, const void* id128) {
    TNN_NEED_INIT();
    TNN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tnn_comm_init: rank %d / world %d", rank, world);
    TNN_REQUIRE(R.comm == nullptr, "tnn_comm_init: communicator already initialised");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    TNN_CHECK_NCCL(R.CommInitRank(&R.comm, world, id, rank));
    R.rank = rank;
    R.world = world;
    return 0;
}

int tnn_comm_destroy(void) {
    if (g_comm_stream) {
        (void)hipStreamSynchronize(g_comm_stream);
        if (g_pending) {
            for (hipEvent_t e : *g_pending) (void)hipEventDestroy(e);
            g_pending->clear();
        }
        if (g_event_pool) {
            for (hipEvent_t e : *g_event_pool) (void)hipEventDestroy(e);
            g_event_pool->clear();
        }
        (void)hipStreamDestroy(g_comm_stream);
        g_comm_stream = nullptr;
    }
    if (R.comm && R.CommDestroy) {
        if (tnn::initialised()) (void)hipStreamSynchronize(tnn::stream());
        R.CommDestroy(R.comm);
    }
    R.comm = nullptr;
    R.rank = 0;
    R.world = 1;
    return 0;
}

int tnn_comm_world(int* rank, int* world) {
    if (!R.comm && tnn::p2p_world(rank, world)) return 0;     // peer-to-peer group without an RCCL communicator
    if (rank) *rank = R.rank;
    if (world) *world = R.world;
    return 0;
}

int tnn_allreduce(void* buf, int64_t n, int dtype, int rop) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    // small f32 sums take the xGMI peer-to-peer path when the peers are mapped (tnn_p2p_connect)
    if (tnn::p2p_can_allreduce(n, dtype, rop)) return tnn::p2p_allreduce((float*)buf, n);
    TNN_REQUIRE(R.comm != nullptr, "tnn_allreduce: tnn_comm_init() has not been called");
    ncclDataType_t t;
    if (int rc = nccl_type(dtype, &t)) return rc;
    ncclRedOp_t op;
    switch (rop) {
        case TNN_RSUM: op = ncclSum; break;
        case TNN_RMAX: op = ncclMax; break;
        case TNN_RMIN: op = ncclMin; break;
        default: tnn::set_error("tnn_allreduce: unknown reduction %d", rop); return 2;
    }
    TNN_CHECK_NCCL(R.AllReduce(buf, buf, (size_t)n, t, op, R.comm, tnn::stream()));
    return 0;
}

int tnn_allreduce_adam(void* grads, int64_t n_reduce, void* p, void* m, void* v, int64_t n_params, double lr,
                       double b1, double b2, double eps, void* pows_f64, int advance, int dtype,
                       int64_t scalar_index, void* scalar_dst) {
    TNN_NEED_INIT();
    TNN_REQUIRE(n_params > 0 && n_reduce >= n_params, "tnn_allreduce_adam: n_reduce (%lld) < n_params (%lld)",
                (long long)n_reduce, (long long)n_params);
    TNN_REQUIRE(pows_f64 != nullptr, "tnn_allreduce_adam: pows state is NULL");
    TNN_REQUIRE(!scalar_dst || (scalar_index >= 0 && scalar_index < n_reduce), "tnn_allreduce_adam: scalar_index");
    const bool aligned = ((reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(p) |
                           reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    // scalar_index inside the vectorised interior is not supported by the fused tail; the trainer's loss slot is
    // the element right behind the parameters
    if (!advance && aligned && (!scalar_dst || scalar_index >= n_params) &&
        tnn::p2p_can_allreduce(n_reduce, dtype, TNN_RSUM))
        return tnn::p2p_allreduce_adam((float*)grads, n_reduce, (float*)p, (float*)m, (float*)v, n_params, lr, b1, b2,
                                       eps, (const double*)pows_f64, scalar_index, (float*)scalar_dst);
    if (int rc = tnn_allreduce(grads, n_reduce, dtype, TNN_RSUM)) return rc;
    const size_t esz = dtype == TNN_F64 ? 8 : 4;
    return tnn_adam_ex(p, grads, m, v, n_params, lr, b1, b2, eps, pows_f64, nullptr, dtype, advance,
                       scalar_dst ? (const char*)grads + (size_t)scalar_index * esz : nullptr, scalar_dst);
}

// Bucketed overlap: a gradient bucket is all-reduced on a separate communication stream as soon as the backward
// launch that produced it has been enqueued, while the library stream goes on with the next layer's backward; the
// optimizer waits for all buckets (tnn_comm_join).  Events carry the two orderings; under hipGraph capture they turn
// into a side branch of the graph.
int tnn_allreduce_async(void* buf, int64_t n, int dtype, int rop) {
    TNN_NEED_INIT();
    if (n <= 0) return 0;
    // nothing to overlap with on the latency path, or no RCCL communicator: the ordinary call
    if (R.comm == nullptr || tnn::p2p_can_allreduce(n, dtype, rop)) return tnn_allreduce(buf, n, dtype, rop);
    ncclDataType_t t;
    if (int rc = nccl_type(dtype, &t)) return rc;
    TNN_REQUIRE(rop == TNN_RSUM, "tnn_allreduce_async: only sums are bucketed");
    if (!g_comm_stream) {
        TNN_CHECK_HIP(hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking));
        g_pending = new std::vector<hipEvent_t>();
    }
    hipEvent_t produced, done;
    if (int rc = take_event(&produced)) return rc;
    if (int rc = take_event(&done)) { give_event(produced); return rc; }
    const auto fail = [&](void) { give_event(produced); give_event(done); return 1; };
    if (hipEventRecord(produced, tnn::stream()) != hipSuccess ||
        hipStreamWaitEvent(g_comm_stream, produced, 0) != hipSuccess) {
        tnn::set_error("tnn_allreduce_async: ordering the communication stream behind the producer failed");
        return fail();
    }
    give_event(produced);
    const ncclResult_t r = R.AllReduce(buf, buf, (size_t)n, t, ncclSum, R.comm, g_comm_stream);
    if (r != ncclSuccess) {
        tnn::set_error("ncclAllReduce (bucket of %lld) -> %s", (long long)n, R.GetErrorString ? R.GetErrorString(r) : "?");
        give_event(done);
        return 1;
    }
    if (hipEventRecord(done, g_comm_stream) != hipSuccess) {
        tnn::set_error("tnn_allreduce_async: recording the bucket-done event failed");
        give_event(done);
        return 1;
    }
    g_pending->push_back(done);
    return 0;
}

int tnn_reduce_scatter(const void* send, void* recv, int64_t n_per_rank, int dtype) {
    TNN_NEED_INIT();
    if (n_per_rank <= 0) return 0;
    const size_t esz = dtype == TNN_F64 || dtype == TNN_I64 ? 8 : dtype == TNN_BF16 ? 2 : dtype == TNN_U8 ? 1 : 4;
    if (R.comm == nullptr) {
        int rank = 0, world = 1;
        (void)tnn_comm_world(&rank, &world);
        // a peer-to-peer group without an RCCL communicator (TNN_COMM=xgmi): the bulk path of tnn_p2p.hip
        if (world > 1 && tnn::p2p_can_bulk(n_per_rank, dtype, true)) return tnn::p2p_reduce_scatter(send, recv, n_per_rank, dtype);
        TNN_REQUIRE(world == 1, "tnn_reduce_scatter: no RCCL communicator (tnn_comm_init) and no peer-to-peer bulk staging "
                                "(tnn_p2p_set_bulk_bytes) that takes %lld elements of dtype %d", (long long)n_per_rank, dtype);
        if (recv != send) TNN_CHECK_HIP(hipMemcpyAsync(recv, send, (size_t)n_per_rank * esz, hipMemcpyDeviceToDevice, tnn::stream()));
        return 0;
    }
    ncclDataType_t t;
    if (int rc = nccl_type(dtype, &t)) return rc;
    TNN_CHECK_NCCL(R.ReduceScatter(send, recv, (size_t)n_per_rank, t, ncclSum, R.comm, tnn::stream()));
    return 0;
}

// A chain = several dependent library calls (reduce-scatter -> optimizer on the owned slice -> all-gather) that run on
// the communication stream while the library stream goes on with the next layer's backward.  While a chain is open,
// tnn::stream() IS the communication stream (tnn::set_stream_override).
static hipEvent_t g_chain_done = nullptr;
static bool g_chain_open = false, g_chain_inline = false;
int tnn_comm_chain_begin(void) {
    TNN_NEED_INIT();
    TNN_REQUIRE(!g_chain_open, "tnn_comm_chain_begin: a chain is already open");
    g_chain_open = true;
    g_chain_inline = R.comm == nullptr;
    if (g_chain_inline) return 0;
    if (!g_comm_stream) {
        TNN_CHECK_HIP(hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking));
        g_pending = new std::vector<hipEvent_t>();
    }
    hipEvent_t produced;
    if (int rc = take_event(&produced)) { g_chain_open = false; return rc; }
    const bool ok = hipEventRecord(produced, tnn::stream()) == hipSuccess &&
                    hipStreamWaitEvent(g_comm_stream, produced, 0) == hipSuccess;
    give_event(produced);
    if (!ok || take_event(&g_chain_done)) {
        g_chain_open = false;
        tnn::set_error("tnn_comm_chain_begin: ordering the communication stream behind the producer failed");
        return 1;
    }
    tnn::set_stream_override(g_comm_stream);
    return 0;
}
int tnn_comm_chain_end(void) {
    TNN_NEED_INIT();
    TNN_REQUIRE(g_chain_open, "tnn_comm_chain_end: no chain is open");
    g_chain_open = false;
    if (g_chain_inline) return 0;
    tnn::set_stream_override(nullptr);
    hipEvent_t done = g_chain_done;
    g_chain_done = nullptr;
    if (hipEventRecord(done, g_comm_stream) != hipSuccess) {
        give_event(done);
        tnn::set_error("tnn_comm_chain_end: recording the chain-done event failed");
        return 1;
    }
    g_pending->push_back(done);
    return 0;
}

int tnn_comm_wait_oldest(void) {
    TNN_NEED_INIT();
    if (!g_pending || g_pending->empty()) return 0;
    hipEvent_t e = g_pending->front();
    g_pending->erase(g_pending->begin());
    const hipError_t st = hipStreamWaitEvent(tnn::stream(), e, 0);
    give_event(e);
    TNN_CHECK_HIP(st);
    return 0;
}

int tnn_comm_join(void) {
    TNN_NEED_INIT();
    if (!g_pending) return 0;
    // also the drain of every error path between the first bucket and the optimizer: nothing stays behind
    hipError_t first_error = hipSuccess;
    for (hipEvent_t e : *g_pending) {
        const hipError_t st = hipStreamWaitEvent(tnn::stream(), e, 0);
        if (first_error == hipSuccess) first_error = st;
        give_event(e);
    }
    g_pending->clear();
    TNN_CHECK_HIP(first_error);
    return 0;
}

int tnn_allgather(const void* send, void* recv, int64_t n_per_rank, int dtype) {
    TNN_NEED_INIT();
    if (n_per_rank <= 0) return 0;
    if (tnn::p2p_can_allgather(n_per_rank, dtype)) return tnn::p2p_allgather(send, recv, n_per_rank, dtype);
    if (R.comm == nullptr) {
        int rank = 0, world = 1;
        (void)tnn_comm_world(&rank, &world);
        if (world > 1 && tnn::p2p_can_bulk(n_per_rank, dtype, false)) return tnn::p2p_allgather_bulk(send, recv, n_per_rank, dtype);
        TNN_REQUIRE(world == 1, "tnn_allgather: no RCCL communicator (tnn_comm_init) and no peer-to-peer bulk staging "
                                "(tnn_p2p_set_bulk_bytes) that takes %lld elements of dtype %d", (long long)n_per_rank, dtype);
        const size_t esz = dtype == TNN_F64 || dtype == TNN_I64 ? 8 : dtype == TNN_BF16 ? 2 : dtype == TNN_U8 ? 1 : 4;
        if (recv != send) TNN_CHECK_HIP(hipMemcpyAsync(recv, send, (size_t)n_per_rank * esz, hipMemcpyDeviceToDevice, tnn::stream()));
        return 0;
    }
    ncclDataType_t t;
    if (int rc = nccl_type(dtype, &t)) return rc;
    TNN_CHECK_NCCL(R.AllGather(send, recv, (size_t)n_per_rank, t, R.comm, tnn::stream()));
    return 0;
}

}  // extern "C"
