// Device plumbing of the C ABI: error string, memory helpers, events.
#include "common.h"

#include <mutex>

namespace srh {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- cache of released device blocks (see common.h).  Best fit within 25 % (+64 KiB) of the request; at most 64
// blocks / 2 GiB are kept, the oldest goes first.  One cache per device.
namespace {
struct Block { void *p; size_t bytes; int dev; };
std::mutex g_pool_mu;
std::vector<Block> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t POOL_MAX_BYTES = (size_t)2 << 30;
constexpr size_t POOL_MAX_BLOCKS = 64;
}  // namespace

void *pool_take(size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); ++i) {
            const Block &b = g_pool[i];
            if (b.dev != dev || b.bytes < bytes || b.bytes > bytes + bytes / 4 + 65536) continue;
            if (best == g_pool.size() || b.bytes < g_pool[best].bytes) best = i;
        }
        if (best != g_pool.size()) {
            void *p = g_pool[best].p;
            g_pool_bytes -= g_pool[best].bytes;
            g_pool.erase(g_pool.begin() + best);
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {                       // make room and try once more
        pool_release();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

void pool_give(void *p, size_t bytes) {
    if (!p) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (bytes > POOL_MAX_BYTES / 2) drop.push_back(p);
        else {
            g_pool.push_back({p, bytes, dev});
            g_pool_bytes += bytes;
            while (g_pool.size() > POOL_MAX_BLOCKS || g_pool_bytes > POOL_MAX_BYTES) {
                drop.push_back(g_pool.front().p);
                g_pool_bytes -= g_pool.front().bytes;
                g_pool.erase(g_pool.begin());
            }
        }
    }
    for (void *q : drop) (void)hipFree(q);
}

void pool_release() {
    std::vector<Block> all;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        all.swap(g_pool);
        g_pool_bytes = 0;
    }
    for (const Block &b : all) (void)hipFree(b.p);
}

}  // namespace srh

extern "C" {

int srh_release_cached(void) {
    srh::pool_release();
    return SRH_OK;
}

const char *srh_last_error(void) { return srh::g_err; }
int srh_version(void) { return 100; }

int srh_device_count(int *count) {
    SRH_REQUIRE(count, "srh_device_count: null argument");
    SRH_CHECK_HIP(hipGetDeviceCount(count));
    return SRH_OK;
}
int srh_set_device(int device) {
    SRH_CHECK_HIP(hipSetDevice(device));
    return SRH_OK;
}
int srh_malloc(void **dptr, size_t bytes) {
    SRH_REQUIRE(dptr, "srh_malloc: null argument");
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) {
        srh::set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return SRH_ENOMEM;
    }
    return SRH_OK;
}
int srh_free(void *dptr) {
    SRH_CHECK_HIP(hipFree(dptr));
    return SRH_OK;
}
int srh_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    SRH_CHECK_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return SRH_OK;
}
int srh_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    SRH_CHECK_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SRH_OK;
}
int srh_memset(void *dptr, int value, size_t bytes) {
    SRH_CHECK_HIP(hipMemset(dptr, value, bytes));
    return SRH_OK;
}
int srh_sync(void) {
    SRH_CHECK_HIP(hipDeviceSynchronize());
    return SRH_OK;
}
int srh_event_create(void **ev) {
    SRH_REQUIRE(ev, "srh_event_create: null argument");
    hipEvent_t e;
    SRH_CHECK_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return SRH_OK;
}
int srh_event_destroy(void *ev) {
    SRH_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return SRH_OK;
}
int srh_event_record(void *ev, void *stream) {
    SRH_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return SRH_OK;
}
int srh_event_elapsed_ms(void *a, void *b, float *ms) {
    SRH_REQUIRE(ms, "srh_event_elapsed_ms: null argument");
    SRH_CHECK_HIP(hipEventSynchronize((hipEvent_t)b));
    SRH_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return SRH_OK;
}

}  // extern "C"
