// runtime.hip -- device runtime of libmi355img: error state, per-device default
// streams, a pooled allocator, copies, events.  This is the part of CuPy the
// reference relies on (cupy.ndarray allocation, memory pool, current stream:
// cupyimg/__init__.py:23-28, _util.py:80) rebuilt as a thin HIP layer.
#include <stdarg.h>

#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"

namespace mi {
std::atomic<int> g_knob_generation{0};


static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    (void)hipGetLastError();
    return (int)e;
}

// ------------------------------------------------------------------ streams
static std::mutex g_stream_mu;
static hipStream_t g_default_stream[64] = {nullptr};

static int default_stream(hipStream_t *out)
{
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index out of range"); return MI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (!g_default_stream[dev]) {
        MI_HIP(hipStreamCreateWithFlags(&g_default_stream[dev], hipStreamNonBlocking));
    }
    *out = g_default_stream[dev];
    return MI_OK;
}

hipStream_t resolve_stream(mi_stream s)
{
    if (s) return (hipStream_t)s;
    hipStream_t d = nullptr;
    if (default_stream(&d) != MI_OK) return nullptr;   // falls back to the null stream
    return d;
}

int current_device_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    return dev & 63;
}

// Compute units of the current device.  One entry per device: a process may drive differently partitioned devices
// (SPX / CPX), and several threads may ask at once (relaxed atomics: every writer stores the same value).
int device_cus()
{
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        hipDeviceProp_t prop;
        n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
        if (n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// ------------------------------------------------------------------ pool
// Size-bucketed caching allocator.  Blocks are rounded up to 512 B below
// 1 MiB and to 2 MiB multiples above; a freed block goes back to its
// (device, size) free list and is reused by the next request of that size.
// Free lists are per stream ("arenas", the contract CuPy's pool gives the
// reference): a block returns to the arena of the stream it was allocated for
// and is only handed out again for work on that stream, so reuse is stream
// ordered without events -- also when callers pass their own streams.
// mi_malloc / mi_free use the arena of the library's default stream.
struct PoolKey {
    int dev;
    size_t size;
    hipStream_t stream;
    bool operator<(const PoolKey &o) const
    {
        if (dev != o.dev) return dev < o.dev;
        if (size != o.size) return size < o.size;
        return stream < o.stream;
    }
};
struct Pool {
    std::mutex mu;
    std::map<PoolKey, std::vector<void *>> free_lists;
    std::unordered_map<void *, PoolKey> live;
    size_t in_use = 0, cached = 0;
};
static Pool g_pool;

static size_t round_size(size_t n)
{
    if (n == 0) n = 1;
    if (n < (1u << 20)) return (n + 511) & ~(size_t)511;
    const size_t g = (size_t)2 << 20;
    return (n + g - 1) / g * g;
}

static int pool_trim_locked()
{
    int keep = 0;
    (void)hipGetDevice(&keep);
    for (auto &kv : g_pool.free_lists) {
        if (kv.second.empty()) continue;
        (void)hipSetDevice(kv.first.dev);
        for (void *p : kv.second) {
            (void)hipFree(p);
            g_pool.cached -= kv.first.size;
        }
        kv.second.clear();
    }
    (void)hipSetDevice(keep);
    return MI_OK;
}

int pool_alloc(void **p, size_t n, hipStream_t stream)
{
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    const size_t sz = round_size(n);
    if (!stream) stream = resolve_stream(nullptr);
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto &fl = g_pool.free_lists[{dev, sz, stream}];
    if (!fl.empty()) {
        *p = fl.back();
        fl.pop_back();
        g_pool.cached -= sz;
    } else {
        hipError_t e = hipMalloc(p, sz);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
            pool_trim_locked();
            e = hipMalloc(p, sz);
        }
        if (e != hipSuccess) {
            *p = nullptr;
            if (e == hipErrorOutOfMemory) {
                (void)hipGetLastError();
                set_error("out of device memory allocating %zu bytes", sz);
                return MI_ERR_NOMEM;
            }
            return hip_fail(e, "hipMalloc");
        }
    }
    g_pool.live[*p] = {dev, sz, stream};
    g_pool.in_use += sz;
    return MI_OK;
}

int pool_free(void *p)
{
    if (!p) return MI_OK;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.live.find(p);
    if (it == g_pool.live.end()) {
        set_error("mi_free: pointer %p was not allocated by mi_malloc", p);
        return MI_ERR_INVALID_ARG;
    }
    g_pool.free_lists[it->second].push_back(p);
    g_pool.in_use -= it->second.size;
    g_pool.cached += it->second.size;
    g_pool.live.erase(it);
    return MI_OK;
}

int Scratch::upload(const void *host, size_t nbytes, hipStream_t stream)
{
    int rc = pool_alloc(&ptr, nbytes, stream);
    if (rc != MI_OK) return rc;
    // pageable source: the runtime stages it before returning, so the caller's
    // buffer may go away; the copy itself is ordered on `stream`.
    MI_HIP(hipMemcpyAsync(ptr, host, nbytes, hipMemcpyHostToDevice, stream));
    return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_version(void) { return MI_VERSION; }
const char *mi_last_error(void) { return g_err; }

int mi_device_count(int *count)
{
    MI_REQUIRE(count, MI_ERR_INVALID_ARG, "count is NULL");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
    return MI_OK;
}
int mi_set_device(int device) { MI_HIP(hipSetDevice(device)); return MI_OK; }
int mi_get_device(int *device) { MI_HIP(hipGetDevice(device)); return MI_OK; }

int mi_device_name(int device, char *buf, size_t buflen)
{
    MI_REQUIRE(buf && buflen > 0, MI_ERR_INVALID_ARG, "bad buffer");
    hipDeviceProp_t prop;
    MI_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return MI_OK;
}

int mi_device_attr(int device, int *cu_count, int *clock_khz, size_t *total_mem)
{
    hipDeviceProp_t prop;
    MI_HIP(hipGetDeviceProperties(&prop, device));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    return MI_OK;
}

int mi_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    size_t f = 0, t = 0;
    MI_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return MI_OK;
}

int mi_malloc(void **ptr, size_t nbytes)
{
    MI_REQUIRE(ptr, MI_ERR_INVALID_ARG, "ptr is NULL");
    return pool_alloc(ptr, nbytes, nullptr);
}
int mi_free(void *ptr) { return pool_free(ptr); }

// test hook (not part of the C-ABI): allocate `nbytes` for work on `stream`, report the block and free it again
int mi_debug_pool_probe(size_t nbytes, mi_stream stream, void **block)
{
    void *p = nullptr;
    int rc = pool_alloc(&p, nbytes, resolve_stream(stream));
    if (rc != MI_OK) return rc;
    if (block) *block = p;
    return pool_free(p);
}

int mi_pool_trim(void)
{
    std::lock_guard<std::mutex> lk(g_pool.mu);
    return pool_trim_locked();
}

int mi_pool_stats(size_t *bytes_in_use, size_t *bytes_cached)
{
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (bytes_in_use) *bytes_in_use = g_pool.in_use;
    if (bytes_cached) *bytes_cached = g_pool.cached;
    return MI_OK;
}

int mi_memcpy_h2d(void *dst, const void *src, size_t nbytes, mi_stream stream)
{
    if (nbytes == 0) return MI_OK;
    MI_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, resolve_stream(stream)));
    return MI_OK;
}

int mi_memcpy_d2h(void *dst, const void *src, size_t nbytes, mi_stream stream)
{
    if (nbytes == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    MI_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    return MI_OK;
}

int mi_memcpy_d2d(void *dst, const void *src, size_t nbytes, mi_stream stream)
{
    if (nbytes == 0) return MI_OK;
    MI_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, resolve_stream(stream)));
    return MI_OK;
}

int mi_memcpy_peer(void *dst, int dst_dev, const void *src, int src_dev, size_t nbytes,
                   mi_stream stream)
{
    if (nbytes == 0) return MI_OK;
    MI_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, nbytes, resolve_stream(stream)));
    return MI_OK;
}

int mi_memset(void *dst, int value, size_t nbytes, mi_stream stream)
{
    if (nbytes == 0) return MI_OK;
    MI_HIP(hipMemsetAsync(dst, value, nbytes, resolve_stream(stream)));
    return MI_OK;
}

int mi_stream_create(mi_stream *stream)
{
    MI_REQUIRE(stream, MI_ERR_INVALID_ARG, "stream is NULL");
    hipStream_t s;
    MI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (mi_stream)s;
    return MI_OK;
}
int mi_stream_destroy(mi_stream stream)
{
    if (stream) MI_HIP(hipStreamDestroy((hipStream_t)stream));
    return MI_OK;
}
int mi_stream_sync(mi_stream stream)
{
    MI_HIP(hipStreamSynchronize(resolve_stream(stream)));
    return MI_OK;
}
int mi_default_stream(mi_stream *stream)
{
    MI_REQUIRE(stream, MI_ERR_INVALID_ARG, "stream is NULL");
    hipStream_t s = nullptr;
    int rc = mi::default_stream(&s);
    *stream = (mi_stream)s;
    return rc;
}
int mi_device_sync(void) { MI_HIP(hipDeviceSynchronize()); return MI_OK; }

int mi_event_create(mi_event *event)
{
    MI_REQUIRE(event, MI_ERR_INVALID_ARG, "event is NULL");
    hipEvent_t e;
    MI_HIP(hipEventCreate(&e));
    *event = (mi_event)e;
    return MI_OK;
}
int mi_event_destroy(mi_event event)
{
    if (event) MI_HIP(hipEventDestroy((hipEvent_t)event));
    return MI_OK;
}
int mi_event_record(mi_event event, mi_stream stream)
{
    MI_HIP(hipEventRecord((hipEvent_t)event, resolve_stream(stream)));
    return MI_OK;
}
int mi_stream_wait_event(mi_stream stream, mi_event event)
{
    MI_HIP(hipStreamWaitEvent(resolve_stream(stream), (hipEvent_t)event, 0));
    return MI_OK;
}
// `producer` / `waiter`: a hipStream_t, NULL = the library's default stream, or the
// __cuda_array_interface__ codes 1 (legacy default stream) / 2 (per-thread default stream)
static hipStream_t foreign_or_own_stream(mi_stream s)
{
    if (s == (mi_stream)(uintptr_t)1) return (hipStream_t) nullptr;   // the null stream IS the legacy default stream
    if (s == (mi_stream)(uintptr_t)2) return hipStreamPerThread;
    return resolve_stream(s);
}
int mi_stream_wait_stream(mi_stream waiter, mi_stream producer)
{
    hipStream_t w = foreign_or_own_stream(waiter), p = foreign_or_own_stream(producer);
    if (w == p) return MI_OK;
    hipEvent_t e;
    MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t err = hipEventRecord(e, p);
    if (err == hipSuccess) err = hipStreamWaitEvent(w, e, 0);
    hipEventDestroy(e);          // released once the recorded work has completed
    MI_HIP(err);
    return MI_OK;
}
int mi_event_sync(mi_event event) { MI_HIP(hipEventSynchronize((hipEvent_t)event)); return MI_OK; }
int mi_event_elapsed_ms(mi_event start, mi_event stop, float *ms)
{
    MI_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MI_OK;
}

}  // extern "C"

/* test / tuning support: a counter that every mi_debug_set_* call advances (common.hpp Knob) */
extern "C" int mi_debug_generation(void) { return mi::g_knob_generation.load(std::memory_order_relaxed); }
