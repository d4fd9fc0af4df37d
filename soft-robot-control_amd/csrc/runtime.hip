// Device plumbing of the C ABI: error string, memory helpers, events.
#include "common.h"

#include <map>
#include <mutex>
#include <utility>

namespace srh {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- cache of released device blocks (see common.h).  hipFree is a DEVICE-WIDE synchronisation (it waits for every
// stream, including a GuSTO request running asynchronously on its own stream: tools/probes/sync_probe.hip), so the
// steady state must never free: requests are rounded up to size classes (quarter octaves above 4 KiB) and a released
// block goes back to the free list of its class, whatever the number of blocks; only when the cache holds more than
// 2 GiB are blocks returned to the driver (largest class first).  One cache per device.
namespace {
std::mutex g_pool_mu;
std::map<std::pair<int, size_t>, std::vector<void *>> g_free;      // (device, class bytes) -> blocks
size_t g_pool_bytes = 0;
constexpr size_t POOL_MAX_BYTES = (size_t)2 << 30;

size_t size_class(size_t bytes) {
    if (bytes <= 4096) return (bytes + 255) & ~(size_t)255;
    size_t p2 = 4096;
    while (p2 * 2 <= bytes) p2 *= 2;                 // p2 <= bytes < 2 p2
    const size_t step = p2 / 4;
    return ((bytes + step - 1) / step) * step;
}
}  // namespace

void *pool_take(size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t cls = size_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_free.find({dev, cls});
        if (it != g_free.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            g_pool_bytes -= cls;
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, cls);
    if (e != hipSuccess) {                       // make room and try once more
        pool_release();
        e = hipMalloc(&p, cls);
    }
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", cls, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

void pool_give(void *p, size_t bytes) {
    if (!p) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t cls = size_class(bytes);
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_free[{dev, cls}].push_back(p);
        g_pool_bytes += cls;
        while (g_pool_bytes > POOL_MAX_BYTES) {          // over budget: the largest cached blocks go back to the driver
            auto big = g_free.end();
            for (auto it = g_free.begin(); it != g_free.end(); ++it)
                if (!it->second.empty() && (big == g_free.end() || it->first.second > big->first.second)) big = it;
            if (big == g_free.end()) break;
            drop.push_back(big->second.back());
            big->second.pop_back();
            g_pool_bytes -= big->first.second;
        }
    }
    for (void *q : drop) (void)hipFree(q);
}

void pool_release() {
    std::vector<void *> all;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (auto &kv : g_free) { all.insert(all.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
        g_pool_bytes = 0;
    }
    for (void *q : all) (void)hipFree(q);
}

}  // namespace srh

extern "C" {

int srh_release_cached(void) {
    srh::pool_release();
    return SRH_OK;
}

const char *srh_last_error(void) { return srh::g_err; }
int srh_version(void) { return 200; }      // major * 100 + minor: 2.0 = the round-2 ABI (async plan solves, sekf_step_projected, sric_dare, spoly_project)

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
int srh_memcpy_d2d(void *dst, const void *src, size_t bytes) {
    SRH_CHECK_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
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
