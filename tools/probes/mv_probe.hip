// What bounds a single-workgroup matrix-vector product with a 350 KB L2-resident matrix?  Variants of the load loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const double cg_t;
__global__ __launch_bounds__(512) void k(double *GT, long long *out, double *sink) {
    __shared__ double uv[512];
    __shared__ double part[512];
    const int tid = threadIdx.x, nm = 400, ldG = 112;
    for (int e = tid; e < nm * ldG; e += 512) GT[e] = 1e-3 * ((e * 7) % 13 - 6);
    uv[tid] = 0.01 * (tid % 11);
    __syncthreads();
    cg_t *G = (cg_t *)GT;
    const int col = tid % 128, grp = tid / 128, cc = col < ldG ? col : ldG - 1;
    long long t0, t1;
    double acc = 0.0;
    for (int rep = 0; rep < 2; ++rep) {
        // A: loads only, 16 per chunk, rows grp + 4 t
        t0 = clock64();
        for (int r0 = grp; r0 < nm; r0 += 64) {
            double gv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { const int r = r0 + 4 * t; gv[t] = G[(size_t)(r < nm ? r : nm - 1) * ldG + cc]; }
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += gv[t];
        }
        __syncthreads();
        t1 = clock64();
        if (tid == 0) out[0] = t1 - t0;
        // B: all 100 loads of the thread at once
        t0 = clock64();
        {
            double gv[100];
#pragma unroll
            for (int t = 0; t < 100; ++t) gv[t] = G[(size_t)(grp + 4 * t) * ldG + cc];
#pragma unroll
            for (int t = 0; t < 100; ++t) acc += gv[t];
        }
        __syncthreads();
        t1 = clock64();
        if (tid == 0) out[1] = t1 - t0;
        // C: 16-byte loads: thread = (column pair, row group of 8)
        t0 = clock64();
        {
            const int cp = tid % 64, g8 = tid / 64;      // 56 column pairs active
            const int c2 = cp < 56 ? cp : 55;
            typedef double v2d __attribute__((ext_vector_type(2))); typedef __attribute__((address_space(1))) const v2d cg2_t;
            cg2_t *G2 = (cg2_t *)GT;
            v2d gv[50];
#pragma unroll
            for (int t = 0; t < 50; ++t) gv[t] = G2[((size_t)(g8 + 8 * t) * ldG) / 2 + c2];
#pragma unroll
            for (int t = 0; t < 50; ++t) acc += gv[t].x + gv[t].y;
        }
        __syncthreads();
        t1 = clock64();
        if (tid == 0) out[2] = t1 - t0;
        // D: as A plus the multiply by uv[r] from LDS
        t0 = clock64();
        for (int r0 = grp; r0 < nm; r0 += 64) {
            double gv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { const int r = r0 + 4 * t; gv[t] = G[(size_t)(r < nm ? r : nm - 1) * ldG + cc]; }
#pragma unroll
            for (int t = 0; t < 16; ++t) { const int r = r0 + 4 * t; acc = fma(gv[t], uv[r < nm ? r : nm - 1], acc); }
        }
        part[tid] = acc;
        __syncthreads();
        t1 = clock64();
        if (tid == 0) out[3] = t1 - t0;
        // E: row-contiguous reading: thread reads 16 B pieces of consecutive memory (pure streaming of the whole matrix)
        t0 = clock64();
        {
            typedef double v2d __attribute__((ext_vector_type(2))); typedef __attribute__((address_space(1))) const v2d cg2_t;
            cg2_t *G2 = (cg2_t *)GT;
            v2d gv[44];                               // 400*112/2 = 22400 double2 / 512 = 43.75
#pragma unroll
            for (int t = 0; t < 44; ++t) { const int e = tid + 512 * t; gv[t] = G2[e < 22400 ? e : 22399]; }
#pragma unroll
            for (int t = 0; t < 44; ++t) acc += gv[t].x + gv[t].y;
        }
        __syncthreads();
        t1 = clock64();
        if (tid == 0) out[4] = t1 - t0;
    }
    sink[tid] = acc + part[(tid + 1) % 512];
}
int main() {
    double *G, *s; long long *o;
    hipMalloc(&G, 400 * 112 * 8); hipMalloc(&s, 512 * 8); hipMalloc(&o, 64);
    k<<<1, 512>>>(G, o, s);
    hipDeviceSynchronize();
    long long out[8];
    hipMemcpy(out, o, 64, hipMemcpyDeviceToHost);
    printf("350 KB from L2, one workgroup (cycles): A 16-load chunks %lld | B 100 loads at once %lld | C 16-byte loads, 50 at once %lld | D chunks + LDS multiply %lld | E contiguous 16-byte stream %lld\n", out[0], out[1], out[2], out[3], out[4]);
    return 0;
}
