// Store-pattern probe for the POD lift: which row-pitched write pattern reaches the HBM write rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// P1: the lift pattern. WG = 4 waves x 32 rows; per column tile of 16: lane (col = l&15, kgrp = l>>4) stores rows kgrp+4*reg (+16)
template <bool NT>
__global__ __launch_bounds__(256) void p_lift(double *out, long B, long n_f, long ldo, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rowbase = (long)blockIdx.x * 128 + wave * 32;
    const int kgrp = lane >> 4;
    const long ntiles = (n_f + 15) / 16;
    const long t0 = (long)blockIdx.y * tiles_per_wg, t1 = min(t0 + tiles_per_wg, ntiles);
    for (long it = t0; it < t1; ++it) {
        const long i = 16 * it + (lane & 15);
        if (i < n_f) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                long r0 = rowbase + kgrp + 4 * reg, r1 = r0 + 16;
                double v = (double)(it + reg);
                if (NT) { __builtin_nontemporal_store(v, &out[r0 * ldo + i]); __builtin_nontemporal_store(v, &out[r1 * ldo + i]); }
                else { out[r0 * ldo + i] = v; out[r1 * ldo + i] = v; }
            }
        }
    }
}
// P3: a wave stores whole 512-byte row segments (64 lanes x 8 B), 32 rows per wave, column blocks of 64
__global__ __launch_bounds__(256) void p_row512(double *out, long B, long n_f, long ldo, int blocks_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rowbase = (long)blockIdx.x * 128 + wave * 32;
    const long nb = (n_f + 63) / 64;
    const long b0 = (long)blockIdx.y * blocks_per_wg, b1 = min(b0 + blocks_per_wg, nb);
    for (long ib = b0; ib < b1; ++ib) {
        const long i = 64 * ib + lane;
        if (i < n_f)
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) out[(rowbase + rr) * ldo + i] = (double)(ib + rr);
    }
}
// P4: 1024-byte row segments (64 lanes x 16 B); needs (row*ldo + i) even -> only for even ldo
__global__ __launch_bounds__(256) void p_row1024(double *out, long B, long n_f, long ldo, int blocks_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rowbase = (long)blockIdx.x * 128 + wave * 32;
    const long nb = (n_f + 127) / 128;
    const long b0 = (long)blockIdx.y * blocks_per_wg, b1 = min(b0 + blocks_per_wg, nb);
    for (long ib = b0; ib < b1; ++ib) {
        const long i = 128 * ib + 2 * lane;
        if (i + 1 < n_f)
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) *reinterpret_cast<d2 *>(&out[(rowbase + rr) * ldo + i]) = d2{(double)ib, (double)rr};
    }
}
// P5: lift pattern, but a wave owns 16 rows and TWO adjacent column tiles per step (256 B per row per instruction pair)
__global__ __launch_bounds__(256) void p_lift_wide(double *out, long B, long n_f, long ldo, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rowbase = (long)blockIdx.x * 64 + wave * 16;
    const int kgrp = lane >> 4;
    const long ntiles = (n_f + 15) / 16;
    const long t0 = (long)blockIdx.y * tiles_per_wg, t1 = min(t0 + tiles_per_wg, ntiles);
    for (long it = t0; it + 1 < t1; it += 2) {
        const long i = 16 * it + (lane & 15);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            long r0 = rowbase + kgrp + 4 * reg;
            double v = (double)(it + reg);
            if (i < n_f) out[r0 * ldo + i] = v;
            if (i + 16 < n_f) out[r0 * ldo + i + 16] = v;
        }
    }
}
__global__ void p_fill(double *out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = 1.0;
}

template <typename F>
static void timeit(const char *name, double bytes, F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    printf("%-40s %.3f ms  %.2f TB/s\n", name, ms / it, bytes / (ms / it * 1e-3) / 1e12);
}

int main() {
    const long B = 65536, n_f = 4884;
    double *out; CK(hipMalloc(&out, sizeof(double) * B * 4928));
    const double bytes = 8.0 * B * n_f;
    timeit("fill linear", bytes, [&] { p_fill<<<4096, 256>>>(out, B * n_f); });
    for (long ldo : {4884L, 4896L, 4864L + 64}) {
        char nm[96];
        for (int ys : {1, 2, 4, 8}) {
            const int tpw = (int)((n_f + 15) / 16 + ys - 1) / ys;
            snprintf(nm, 96, "lift pattern ldo=%ld ysplit=%d", ldo, ys);
            timeit(nm, bytes, [&] { p_lift<false><<<dim3(B / 128, ys), 256>>>(out, B, n_f, ldo, tpw); });
        }
        snprintf(nm, 96, "lift pattern nt ldo=%ld ysplit=2", ldo);
        timeit(nm, bytes, [&] { p_lift<true><<<dim3(B / 128, 2), 256>>>(out, B, n_f, ldo, (int)((n_f + 15) / 16 + 1) / 2); });
        snprintf(nm, 96, "row512 ldo=%ld ysplit=2", ldo);
        timeit(nm, bytes, [&] { p_row512<<<dim3(B / 128, 2), 256>>>(out, B, n_f, ldo, (int)((n_f + 63) / 64 + 1) / 2); });
        snprintf(nm, 96, "row1024 ldo=%ld ysplit=2", ldo);
        timeit(nm, bytes, [&] { p_row1024<<<dim3(B / 128, 2), 256>>>(out, B, n_f, ldo, (int)((n_f + 127) / 128 + 1) / 2); });
        snprintf(nm, 96, "lift wide (16 rows x 32 cols) ldo=%ld", ldo);
        timeit(nm, bytes, [&] { p_lift_wide<<<dim3(B / 64, 2), 256>>>(out, B, n_f, ldo, (int)((n_f + 15) / 16 + 1) / 2); });
    }
    return 0;
}
