// Single-workgroup issue / latency probe for gfx950: what one wave (and 8 waves on one CU) pay per dependent f64 FMA,
// per LDS round trip, per barrier, per v_readlane broadcast.  clock64() units against wall_clock64() (100 MHz).
// Build: hipcc -O3 --offload-arch=gfx950 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) void probe(double *out, long long *t, int iters, double seed) {
    __shared__ double sm[1024];
    const int tid = threadIdx.x;
    sm[tid] = seed + tid; sm[tid + 512] = seed;
    __syncthreads();
    long long c0, c1; 
    double x = seed, y = seed * 0.5, z = 1.0;
    // (0) dependent FMA chain, all 8 waves
    c0 = clock64(); long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) { x = fma(x, y, z); x = fma(x, y, z); x = fma(x, y, z); x = fma(x, y, z); }
    c1 = clock64(); long long w1 = wall_clock64();
    if (tid == 0) { t[0] = c1 - c0; t[10] = w1 - w0; }
    // (1) dependent FMA chain, wave 0 only
    __syncthreads();
    c0 = clock64();
    if (tid < 64) for (int i = 0; i < iters; ++i) { x = fma(x, y, z); x = fma(x, y, z); x = fma(x, y, z); x = fma(x, y, z); }
    c1 = clock64();
    if (tid == 0) t[1] = c1 - c0;
    // (2) 4 independent FMA chains, wave 0 only
    __syncthreads();
    double a = x, b = x + 1, c = x + 2, d = x + 3;
    c0 = clock64();
    if (tid < 64) for (int i = 0; i < iters; ++i) { a = fma(a, y, z); b = fma(b, y, z); c = fma(c, y, z); d = fma(d, y, z); }
    c1 = clock64();
    if (tid == 0) t[2] = c1 - c0;
    x = a + b + c + d;
    // (3) LDS dependent round trip (pointer chase), wave 0
    __syncthreads();
    int idx = tid & 63;
    c0 = clock64();
    if (tid < 64) for (int i = 0; i < iters; ++i) { idx = (int)sm[idx] & 63; idx = (int)sm[idx + 64] & 63; idx = (int)sm[idx + 128] & 63; idx = (int)sm[idx + 192] & 63; }
    c1 = clock64();
    if (tid == 0) t[3] = c1 - c0;
    x += idx;
    // (4) barriers, all waves
    __syncthreads();
    c0 = clock64();
    for (int i = 0; i < iters; ++i) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
    c1 = clock64();
    if (tid == 0) t[4] = c1 - c0;
    // (5) readlane + fma, wave 0
    c0 = clock64();
    if (tid < 64) for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(y), k + 1), hi = __builtin_amdgcn_readlane(__double2hiint(y), k + 1);
            x = fma(x, __hiloint2double(hi, lo), z);
        }
    }
    c1 = clock64();
    if (tid == 0) t[5] = c1 - c0;
    // (6) rcp + 2 newton, wave 0, dependent
    __syncthreads();
    c0 = clock64();
    if (tid < 64) for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); r = fma(fma(-x, r, 1.0), r, r); x = r + 1.5; }
    }
    c1 = clock64();
    if (tid == 0) t[6] = c1 - c0;
    // (7) IEEE division, wave 0, dependent
    __syncthreads();
    c0 = clock64();
    if (tid < 64) for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x = 1.0 / x + 1.5;
    }
    c1 = clock64();
    if (tid == 0) t[7] = c1 - c0;
    // (8) LDS write -> barrier -> broadcast read -> fma (one elimination-step skeleton), all waves
    __syncthreads();
    c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (tid == k) sm[512 + k] = x;
            __syncthreads();
            x = fma(x, sm[512 + k], z);
        }
    }
    c1 = clock64();
    if (tid == 0) t[8] = c1 - c0;
    // (9) sqrt, wave 0, dependent
    __syncthreads();
    c0 = clock64();
    if (tid < 64) for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x = sqrt(x) + 1.5;
    }
    c1 = clock64();
    if (tid == 0) t[9] = c1 - c0;
    out[tid] = x + y + z;
}

int main() {
    double *out; long long *t;
    hipMalloc(&out, 512 * 8); hipMalloc(&t, 16 * 8);
    const int iters = 256;
    for (int rep = 0; rep < 2; ++rep) {
        probe<<<1, 512>>>(out, t, iters, 1.0000001);
        hipDeviceSynchronize();
    }
    long long h[16];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"dependent fma, 8 waves", "dependent fma, 1 wave", "4 independent fma chains, 1 wave", "LDS pointer chase, 1 wave",
                           "barrier, 8 waves", "readlane pair + fma, 1 wave", "rcp + 2 newton (+add), 1 wave", "IEEE div (+add), 1 wave",
                           "lds write/barrier/broadcast read/fma, 8 waves", "sqrt (+add), 1 wave"};
    printf("clock64 per wall_clock64 tick (100 MHz): %.2f  -> %.0f MHz\n", (double)h[0] / h[10], 100.0 * h[0] / h[10]);
    for (int i = 0; i < 10; ++i) printf("%-48s %8.1f clocks per op\n", names[i], (double)h[i] / (4.0 * iters));
    return 0;
}
