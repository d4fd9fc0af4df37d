// Sustained v_mfma_f64_16x16x4 rate of the WHOLE chip (every CU busy), 1 / 2 / 4 waves per SIMD: the matrix-pipe
// ceiling a balanced kernel (POD projection: 8 flop per byte) has to be priced against, next to the HBM ceiling.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_rate_probe.hip -o bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void rate(double *out, int iters) {
    const int l = threadIdx.x & 63;
    double a = l * 0.5, b = l * 0.25;
    d4 c[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) c[k] = d4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < CH; ++k) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) s += c[k][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
static void run(double *out, int wgs, int threads, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<CH><<<wgs, threads>>>(out, iters);
    hipEventRecord(e0);
    rate<CH><<<wgs, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)wgs * (threads / 64) * iters * CH;      // MFMA instructions
    const double per_simd = (double)iters * CH * (threads / 64) * wgs / (256.0 * 4);
    printf("wgs %4d x %4d threads, %d chains: %7.1f us  %6.1f TFLOP/s  %5.1f ns per MFMA per SIMD\n", wgs, threads, CH, ms * 1e3,
           n * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / per_simd);
}

int main() {
    double *out; hipMalloc(&out, 8 << 20);
    run<1>(out, 1, 64, 4000);
    run<4>(out, 1, 64, 1000);
    run<1>(out, 256, 256, 4000);
    run<4>(out, 256, 256, 1000);
    run<1>(out, 256, 512, 4000);
    run<2>(out, 256, 512, 2000);
    run<4>(out, 256, 512, 1000);
    run<4>(out, 256, 1024, 1000);
    run<4>(out, 1024, 256, 1000);
    return 0;
}
