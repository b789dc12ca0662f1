// Runtime of libtnn_hip.so: device/stream ownership, caching pool allocator, copies, events,
// hipGraph capture.  Replaces what numpy's allocator does behind every ndarray the reference creates
// (core/tensor.py:20 np.asarray, :171 np.zeros) with HBM buffers that are recycled without touching
// hipMalloc on the hot path.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "tnn_internal.h"

namespace {

thread_local char g_err[1024] = "";

struct Block {
    size_t size;       // rounded (bucket) size
    uint64_t graph;    // 0 = general pool, else id of the graph that owns the buffer
};

struct GraphRec {
    uint64_t id = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::multimap<size_t, void*> free_list;   // buffers owned by this graph, currently unused
};

struct State {
    bool ready = false;
    int device = 0;
    int cus = 256;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::unordered_map<void*, Block> blocks;          // every buffer handed out by hipMalloc
    std::multimap<size_t, void*> free_list;           // general pool
    int64_t live_bytes = 0, cached_bytes = 0, device_allocs = 0;
    // capture
    uint64_t next_graph_id = 1;
    GraphRec* capturing = nullptr;
    std::unordered_map<uint64_t, GraphRec*> graphs;
    std::vector<GraphRec*> destroy_later;             // graphs released while another capture was open (tnn_graph_destroy)
    int* fault = nullptr;                             // host-pinned, device-visible sticky fault word (tnn::fault_word)
};
State g;

// bucket sizes: 512 B granules up to 64 KiB, then 1/8-of-power-of-two steps (<= 12.5 % slack)
size_t bucket(size_t bytes) {
    if (bytes == 0) bytes = 1;
    if (bytes <= 65536) return (bytes + 511) & ~size_t(511);
    size_t p = 1;
    while (p < bytes) p <<= 1;
    size_t step = p >> 4;   // p/2 < bytes <= p ; steps of p/16 over the upper half
    return (bytes + step - 1) / step * step;
}

}  // namespace

namespace tnn {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local hipStream_t g_stream_override = nullptr;
hipStream_t stream() { return g_stream_override ? g_stream_override : g.stream; }
void set_stream_override(hipStream_t s) { g_stream_override = s; }
bool initialised() { return g.ready; }
int num_cus() { return g.cus; }

int* fault_word() { return g.fault; }
// Sticky device fault (a kernel's bounded in-launch wait ran out, tnn_internal.h TNN_FAULT_*): reported by every
// synchronising entry point from then on — results produced after the fault are not to be trusted.
int check_fault(const char* where) {
    const int code = g.fault ? __atomic_load_n(g.fault, __ATOMIC_RELAXED) : 0;
    if (code == 0) return 0;
    set_error("%s: a kernel reported device fault %d (%s); outputs since then are invalid", where, code,
              code == TNN_FAULT_SPLITK_HANDOFF ? "the split-K bf16 GEMM's partner workgroup did not publish its slab in time" : "unknown");
    return 1;
}

static const int* g_update_guard = nullptr;
const int* update_guard() { return g_update_guard; }
void set_update_guard(const int* device_word) { g_update_guard = device_word; }
}  // namespace tnn

extern "C" {

const char* tnn_last_error(void) { return g_err; }
int tnn_backend_kind(void) { return 1; }

int tnn_init(int device) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ready) {
        TNN_REQUIRE(device == g.device, "tnn_init: already bound to device %d", g.device);
        return 0;
    }
    int n = 0;
    TNN_CHECK_HIP(hipGetDeviceCount(&n));
    TNN_REQUIRE(n > 0, "tnn_init: no HIP device visible");
    TNN_REQUIRE(device >= 0 && device < n, "tnn_init: device %d out of range (%d visible)", device, n);
    TNN_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t p;
    TNN_CHECK_HIP(hipGetDeviceProperties(&p, device));
    g.cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    TNN_CHECK_HIP(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    TNN_CHECK_HIP(hipHostMalloc((void**)&g.fault, 64, hipHostMallocMapped));
    *g.fault = 0;
    g.device = device;
    g.ready = true;
    return 0;
}

int tnn_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ready) return 0;
    hipStreamSynchronize(g.stream);
    for (auto& kv : g.graphs) {
        if (kv.second->exec) hipGraphExecDestroy(kv.second->exec);
        if (kv.second->graph) hipGraphDestroy(kv.second->graph);
        delete kv.second;
    }
    g.graphs.clear();
    for (auto& kv : g.blocks) hipFree(kv.first);
    g.blocks.clear();
    g.free_list.clear();
    g.live_bytes = g.cached_bytes = 0;
    hipStreamDestroy(g.stream);
    g.stream = nullptr;
    if (g.fault) hipHostFree(g.fault);
    g.fault = nullptr;
    g.ready = false;
    return 0;
}

int tnn_device_props(int* cu_count, int* clock_khz, int64_t* hbm_bytes, char* name, int name_len) {
    TNN_NEED_INIT();
    hipDeviceProp_t p;
    TNN_CHECK_HIP(hipGetDeviceProperties(&p, g.device));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return 0;
}

int tnn_malloc(size_t bytes, void** out) {
    TNN_NEED_INIT();
    TNN_REQUIRE(out != nullptr, "tnn_malloc: out is NULL");
    size_t sz = bucket(bytes);
    std::lock_guard<std::mutex> lk(g.mu);
    void* p = nullptr;
    if (g.capturing) {
        auto it = g.capturing->free_list.find(sz);
        if (it != g.capturing->free_list.end()) {
            p = it->second;
            g.capturing->free_list.erase(it);
        }
    }
    if (!p) {
        auto it = g.free_list.find(sz);
        if (it != g.free_list.end()) {
            p = it->second;
            g.free_list.erase(it);
            g.cached_bytes -= (int64_t)sz;
        }
    }
    if (!p) {
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            // give cached buffers back to the driver once, then retry
            (void)hipGetLastError();
            hipStreamSynchronize(g.stream);
            for (auto& kv : g.free_list) {
                hipFree(kv.second);
                g.blocks.erase(kv.second);
            }
            g.free_list.clear();
            g.cached_bytes = 0;
            e = hipMalloc(&p, sz);
        }
        if (e != hipSuccess) {
            tnn::set_error("tnn_malloc: hipMalloc(%zu) -> %s", sz, hipGetErrorString(e));
            return 1;
        }
        g.device_allocs++;
        g.blocks[p] = Block{sz, 0};
    }
    g.blocks[p].graph = g.capturing ? g.capturing->id : 0;
    g.live_bytes += (int64_t)sz;
    *out = p;
    return 0;
}

int tnn_free(void* p) {
    if (!p) return 0;
    if (!g.ready) return 0;   // interpreter teardown after tnn_shutdown: buffers already released
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g.blocks.find(p);
    TNN_REQUIRE(it != g.blocks.end(), "tnn_free: %p was not allocated by tnn_malloc", p);
    g.live_bytes -= (int64_t)it->second.size;
    uint64_t gid = it->second.graph;
    if (gid != 0) {
        auto gi = g.graphs.find(gid);
        if (gi != g.graphs.end()) {   // owned by a live graph: never recycled outside it
            gi->second->free_list.emplace(it->second.size, p);
            return 0;
        }
        it->second.graph = 0;
    }
    g.free_list.emplace(it->second.size, p);
    g.cached_bytes += (int64_t)it->second.size;
    return 0;
}

int tnn_pool_stats(int64_t* live_bytes, int64_t* cached_bytes, int64_t* device_allocs) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (live_bytes) *live_bytes = g.live_bytes;
    if (cached_bytes) *cached_bytes = g.cached_bytes;
    if (device_allocs) *device_allocs = g.device_allocs;
    return 0;
}

int tnn_pool_trim(void) {
    TNN_NEED_INIT();
    TNN_CHECK_HIP(hipStreamSynchronize(g.stream));
    std::lock_guard<std::mutex> lk(g.mu);
    for (auto& kv : g.free_list) {
        hipFree(kv.second);
        g.blocks.erase(kv.second);
    }
    g.free_list.clear();
    g.cached_bytes = 0;
    return 0;
}

int tnn_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    TNN_NEED_INIT();
    if (bytes == 0) return 0;
    TNN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g.stream));
    TNN_CHECK_HIP(hipStreamSynchronize(g.stream));
    return 0;
}

int tnn_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    TNN_NEED_INIT();
    if (bytes == 0) return 0;
    TNN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g.stream));
    TNN_CHECK_HIP(hipStreamSynchronize(g.stream));
    return tnn::check_fault("tnn_memcpy_d2h");
}

int tnn_memcpy_d2d(void* dst, const void* src, size_t bytes) {
    TNN_NEED_INIT();
    if (bytes == 0) return 0;
    TNN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g.stream));
    return 0;
}

int tnn_memset(void* dst, int byte, size_t bytes) {
    TNN_NEED_INIT();
    if (bytes == 0) return 0;
    TNN_CHECK_HIP(hipMemsetAsync(dst, byte, bytes, g.stream));
    return 0;
}

int tnn_stream_sync(void) {
    TNN_NEED_INIT();
    TNN_CHECK_HIP(hipStreamSynchronize(g.stream));
    return tnn::check_fault("tnn_stream_sync");
}

int tnn_event_create(void** ev) {
    TNN_NEED_INIT();
    hipEvent_t e;
    TNN_CHECK_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return 0;
}
int tnn_event_record(void* ev) {
    TNN_NEED_INIT();
    TNN_CHECK_HIP(hipEventRecord((hipEvent_t)ev, g.stream));
    return 0;
}
int tnn_event_elapsed_ms(void* start, void* stop, float* ms) {
    TNN_NEED_INIT();
    TNN_CHECK_HIP(hipEventSynchronize((hipEvent_t)stop));
    TNN_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}
int tnn_event_destroy(void* ev) {
    if (ev) hipEventDestroy((hipEvent_t)ev);
    return 0;
}

static int graph_destroy_now(GraphRec* r);

int tnn_graph_capture_begin(void) {
    TNN_NEED_INIT();
    std::lock_guard<std::mutex> lk(g.mu);
    TNN_REQUIRE(g.capturing == nullptr, "tnn_graph_capture_begin: a capture is already open");
    GraphRec* r = new GraphRec();
    r->id = g.next_graph_id++;
    hipError_t e = hipStreamBeginCapture(g.stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        delete r;
        tnn::set_error("hipStreamBeginCapture -> %s", hipGetErrorString(e));
        return 1;
    }
    g.graphs[r->id] = r;
    g.capturing = r;
    return 0;
}

int tnn_graph_capture_end(void** graph_exec) {
    TNN_NEED_INIT();
    GraphRec* r;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        TNN_REQUIRE(g.capturing != nullptr, "tnn_graph_capture_end: no capture is open");
        r = g.capturing;
        g.capturing = nullptr;
    }
    const hipError_t e_end = hipStreamEndCapture(g.stream, &r->graph);
    std::vector<GraphRec*> later;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        later.swap(g.destroy_later);
    }
    for (GraphRec* d : later) graph_destroy_now(d);           // graphs whose owners died while this capture was open
    TNN_CHECK_HIP(e_end);
    TNN_CHECK_HIP(hipGraphInstantiate(&r->exec, r->graph, nullptr, nullptr, 0));
    *graph_exec = (void*)r;
    return 0;
}

int tnn_graph_launch(void* graph_exec) {
    TNN_NEED_INIT();
    GraphRec* r = (GraphRec*)graph_exec;
    TNN_REQUIRE(r && r->exec, "tnn_graph_launch: invalid graph");
    TNN_CHECK_HIP(hipGraphLaunch(r->exec, g.stream));
    return 0;
}

int tnn_graph_destroy(void* graph_exec) {
    if (!graph_exec || !g.ready) return 0;
    GraphRec* r = (GraphRec*)graph_exec;
    {
        // While ANOTHER capture is open (a garbage-collected graph object dying in the middle of a re-capture) the stream
        // must not be synchronised — that invalidates the open capture: the destruction waits for tnn_graph_capture_end.
        std::lock_guard<std::mutex> lk(g.mu);
        if (g.capturing != nullptr && g.capturing != r) {
            g.destroy_later.push_back(r);
            return 0;
        }
    }
    return graph_destroy_now(r);
}

static int graph_destroy_now(GraphRec* r) {
    hipStreamSynchronize(g.stream);
    std::lock_guard<std::mutex> lk(g.mu);
    if (r->exec) hipGraphExecDestroy(r->exec);
    if (r->graph) hipGraphDestroy(r->graph);
    // buffers the graph owned go back to the general pool; still-live ones lose their tag
    for (auto& kv : r->free_list) {
        g.free_list.emplace(kv.first, kv.second);
        g.cached_bytes += (int64_t)kv.first;
        g.blocks[kv.second].graph = 0;
    }
    for (auto& kv : g.blocks)
        if (kv.second.graph == r->id) kv.second.graph = 0;
    g.graphs.erase(r->id);
    delete r;
    return 0;
}

}  // extern "C"
