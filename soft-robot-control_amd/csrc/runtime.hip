// Device plumbing of the C ABI: error string, memory helpers, events.
#include "common.h"

namespace srh {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace srh

extern "C" {

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
