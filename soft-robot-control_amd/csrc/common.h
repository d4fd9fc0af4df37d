// Shared helpers for libsofacontrol_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/sofacontrol_hip.h"

namespace srh {

void set_error(const char *fmt, ...);

#define SRH_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            srh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                           __LINE__);                                                         \
            return SRH_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define SRH_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            srh::set_error(__VA_ARGS__);      \
            return SRH_EINVAL;                \
        }                                     \
    } while (0)

// Device allocations of the host-pointer entry points come from a small cache of released blocks (runtime.hip): a call
// like silqr_solve makes ~15 allocations, and hipMalloc / hipFree (the latter a device-wide synchronisation) cost more
// than the kernels of a small batch.  Blocks are only handed back after the call that used them has synchronised.
void *pool_take(size_t bytes);               // nullptr on failure (error string set)
void pool_give(void *p, size_t bytes);
void pool_release();                         // hipFree everything that is cached

// RAII device buffer used by the host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) pool_give(p, bytes); }
    int alloc(size_t n) {
        if (p) { pool_give(p, bytes); p = nullptr; }
        bytes = n;
        if (n == 0) return SRH_OK;
        p = pool_take(n);
        return p ? SRH_OK : SRH_ENOMEM;
    }
    int upload(const void *src, size_t n) {
        int rc = alloc(n);
        if (rc) return rc;
        if (n == 0) return SRH_OK;
        hipError_t e = hipMemcpy(p, src, n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_error("hipMemcpy H2D failed: %s", hipGetErrorString(e)); return SRH_EHIP; }
        return SRH_OK;
    }
    int download(void *dst, size_t n) const {
        if (n == 0) return SRH_OK;
        hipError_t e = hipMemcpy(dst, p, n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("hipMemcpy D2H failed: %s", hipGetErrorString(e)); return SRH_EHIP; }
        return SRH_OK;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Dynamic LDS of a launch, as it should be REQUESTED: whole 2560-byte units (capped at the 160 KB of a CU).  The 160 KB LDS of
// gfx950 is allocated in 1280-byte granules; a 161 568-byte request (not a multiple) was observed to end at 161 280 bytes --
// the last ints of the kernel's carve read back as zeros and their stores were dropped (round 3, lean SCP kernels).
inline size_t lds_request(size_t bytes) {
    const size_t r = (bytes + 2559) / 2560 * 2560;
    return r < (size_t)160 * 1024 ? r : (bytes > (size_t)160 * 1024 ? bytes : (size_t)160 * 1024);
}

}  // namespace srh
