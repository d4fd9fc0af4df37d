// How the shape of a workgroup's read tile changes the HBM rate of a row-major (B x n_f) stream (pitch 8 n_f bytes,
// never a multiple of 128): every workgroup walks along K over its R rows in chunks of SEG bytes per row, NF chunks in
// flight per thread (8 x 16-byte loads each).  occupancy is pinned with dynamic LDS, as a staged kernel would have it.
// Build: hipcc -O3 --offload-arch=gfx950 tile_read_probe.hip -o bin/tile_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));

// LPR = lanes per row segment (SEG = 16 LPR bytes); a thread's 8 loads of a chunk cover rows r0 + (256 / LPR) u
template <int LPR, int NF>
__global__ __launch_bounds__(256) void tile_kernel(const double *__restrict__ X, long B, long n_f, double *out) {
    extern __shared__ char smem[];
    constexpr int RPP = 256 / LPR;        // rows per pass of the 256 threads
    constexpr int R = 8 * RPP;            // rows per workgroup
    constexpr int SEGD = 2 * LPR;         // doubles per row per chunk
    const int t = threadIdx.x;
    const long row0 = (long)blockIdx.x * R + t / LPR;
    const int piece = t % LPR;
    const long nch = n_f / SEGD;
    double s = 0.0;
    const double *base = X + row0 * n_f + 2 * piece;
    long c = 0;
    for (; c + NF <= nch; c += NF) {
        d2 v[NF][8];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[f][u] = __builtin_nontemporal_load((const d2 *)(base + (long)u * RPP * n_f + (c + f) * SEGD));
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[f][u].x + v[f][u].y;
    }
    if (s == 1.2345e300) out[0] = s + smem[0];
}

template <int LPR, int NF>
static void run(const double *X, long B, long n_f, double *out, size_t lds) {
    constexpr int R = 8 * (256 / LPR);
    hipFuncSetAttribute((const void *)tile_kernel<LPR, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = (int)(B / R);
    for (int w = 0; w < 2; ++w) tile_kernel<LPR, NF><<<wgs, 256, lds>>>(X, B, n_f, out);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) tile_kernel<LPR, NF><<<wgs, 256, lds>>>(X, B, n_f, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)B * (n_f / (2 * LPR)) * (2 * LPR) * 8;
    printf("rows/WG %4d  seg %5d B  chunks in flight %d  LDS %3zu KB (WG/CU pinned) : %7.1f us  %6.0f GB/s\n", R, 16 * LPR, NF,
           lds >> 10, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
}

// the projection kernel's own shape: a load instruction covers 16 rows x 64 bytes per wave (lane = row, k-group), the 8
// instructions of a chunk walk along the same rows (512 bytes per row and chunk)
template <int NF>
__global__ __launch_bounds__(256) void mfma_shape_kernel(const double *__restrict__ X, long B, long n_f, double *out) {
    extern __shared__ char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long row = (long)blockIdx.x * 64 + wave * 16 + (lane & 15);
    const double *base = X + row * n_f + 2 * (lane >> 4);
    const long nch = n_f / 64;
    double s = 0.0;
    long c = 0;
    for (; c + NF <= nch; c += NF) {
        d2 v[NF][8];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[f][q] = *(const d2 *)(base + (c + f) * 64 + 8 * q);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[f][q].x + v[f][q].y;
    }
    if (s == 1.2345e300) out[0] = s + smem[0];
}

template <int NF>
static void run_mfma_shape(const double *X, long B, long n_f, double *out, size_t lds) {
    hipFuncSetAttribute((const void *)mfma_shape_kernel<NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = (int)(B / 64);
    for (int w = 0; w < 2; ++w) mfma_shape_kernel<NF><<<wgs, 256, lds>>>(X, B, n_f, out);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) mfma_shape_kernel<NF><<<wgs, 256, lds>>>(X, B, n_f, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)B * (n_f / 64) * 64 * 8;
    printf("MFMA-operand shape (16 rows x 64 B per instruction, 512 B per row and chunk), chunks in flight %d, LDS %3zu KB: %7.1f us  %6.0f GB/s\n",
           NF, lds >> 10, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
}

int main() {
    const long B = 65536, n_f = 4884;
    double *X, *out;
    hipMalloc(&X, (size_t)B * n_f * 8); hipMalloc(&out, 64);
    hipMemset(X, 0, (size_t)B * n_f * 8);
    for (size_t lds : {(size_t)150 << 10, (size_t)75 << 10, (size_t)36 << 10}) {
        run<4, 1>(X, B, n_f, out, lds);  run<4, 2>(X, B, n_f, out, lds);
        run<16, 1>(X, B, n_f, out, lds); run<16, 2>(X, B, n_f, out, lds);
        run<32, 1>(X, B, n_f, out, lds); run<32, 2>(X, B, n_f, out, lds);
        run<64, 1>(X, B, n_f, out, lds); run<64, 2>(X, B, n_f, out, lds);
        run<32, 4>(X, B, n_f, out, lds); run<64, 4>(X, B, n_f, out, lds);
    }
    for (size_t lds : {(size_t)75 << 10, (size_t)36 << 10, (size_t)16 << 10}) {
        run_mfma_shape<1>(X, B, n_f, out, lds); run_mfma_shape<2>(X, B, n_f, out, lds); run_mfma_shape<4>(X, B, n_f, out, lds);
    }
    return 0;
}
