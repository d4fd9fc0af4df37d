// Which runtime calls wait for a kernel running on ANOTHER, non-blocking stream?  (ROCm 7.2, gfx950)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int *out) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (out) *out = 1;
}
__global__ void tiny(int *p) { if (p) p[threadIdx.x] = threadIdx.x; }
static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d = nullptr, *d2 = nullptr;
    hipMalloc(&d, 4096); hipMalloc(&d2, 4096);
    std::vector<int> h(1024, 0);
    int *pin = nullptr; hipHostMalloc((void **)&pin, 4096, hipHostMallocDefault);
    tiny<<<1, 64>>>(d); hipDeviceSynchronize();
    const char *names[] = {"hipMemcpy H2D pageable", "hipMemcpy D2H pageable", "hipMalloc (1 MiB)", "hipMemcpyAsync(null) pageable H2D + sync(null)",
                           "hipMemcpyAsync(null) pinned D2H + sync(null)", "kernel(null) + hipStreamSynchronize(null)", "hipMemcpy D2H pinned", "hipFree (1 MiB)", "hipHostMalloc (64 KiB)", "hipStreamCreate", "hipEventCreate + hipEventRecord(null)"};
    void *keep = nullptr; void *hp = nullptr; hipStream_t s2; hipEvent_t ev;
    for (int v = 0; v < 11; ++v) {
        spin<<<1, 64, 0, s>>>(40000000LL, d2);        // ~20 ms at 2 GHz
        auto t0 = std::chrono::steady_clock::now();
        switch (v) {
            case 0: hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice); break;
            case 1: hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost); break;
            case 2: { hipMalloc(&keep, 1 << 20); break; }
            case 3: hipMemcpyAsync(d, h.data(), 256, hipMemcpyHostToDevice, nullptr); hipStreamSynchronize(nullptr); break;
            case 4: hipMemcpyAsync(pin, d, 256, hipMemcpyDeviceToHost, nullptr); hipStreamSynchronize(nullptr); break;
            case 5: tiny<<<1, 64>>>(d); hipStreamSynchronize(nullptr); break;
            case 6: hipMemcpy(pin, d, 256, hipMemcpyDeviceToHost); break;
            case 7: hipFree(keep); break;
            case 8: hipHostMalloc(&hp, 65536, hipHostMallocDefault); break;
            case 9: hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); break;
            case 10: hipEventCreate(&ev); hipEventRecord(ev, nullptr); break;
        }
        const double t_call = ms_since(t0);
        hipStreamSynchronize(s);
        printf("%-52s %8.3f ms (spin kernel done after %8.3f ms)\n", names[v], t_call, ms_since(t0));
    }
    return 0;
}
