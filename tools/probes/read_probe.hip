// What a pure read stream reaches on this GPU: grid-stride sum over a resident buffer with U 16-byte loads in flight per
// thread.  The ceiling any one-pass HBM-bound kernel of this repo can be held against (roofline notes).
// Build: hipcc -O3 --offload-arch=gfx950 read_probe.hip -o bin/read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const d2 *__restrict__ p, size_t n2, double *out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) { s0 += v[u].x; s1 += v[u].y; }
    }
    for (; i < n2; i += stride) { d2 v = p[i]; s0 += v.x; s1 += v.y; }
    if (s0 + s1 == 1.2345e300) out[0] = s0;      // keep the loads
}

template <int U>
static void run(const d2 *p, size_t bytes, int wgs, double *out, const char *tag) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) read_kernel<U><<<wgs, 256>>>(p, bytes / 16, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) read_kernel<U><<<wgs, 256>>>(p, bytes / 16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-8s %6.0f MB  wgs %5d  U %2d : %7.1f us  %6.0f GB/s\n", tag, bytes / 1e6, wgs, U, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
}

int main() {
    const size_t big = (size_t)2560 << 20;
    d2 *p; double *out;
    hipMalloc(&p, big); hipMalloc(&out, 64);
    hipMemset(p, 0, big);
    const size_t sizes[] = {(size_t)4884 * 4884 * 8, big};
    for (size_t bytes : sizes) {
        for (int wgs : {256, 512, 1024, 2048, 4096, 8192}) {
            run<4>(p, bytes, wgs, out, "read");
            run<8>(p, bytes, wgs, out, "read");
            run<16>(p, bytes, wgs, out, "read");
        }
    }
    return 0;
}
