// Layout and issue rate of v_mfma_f64_4x4x4 (4 blocks) next to v_mfma_f64_16x16x4 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void one(const double *a, const double *b, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <int KIND>
__global__ void rate(double *out, int iters) {
    const int l = threadIdx.x & 63;
    double a = l * 0.5, b = l * 0.25;
    if (KIND == 0) {
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
    }
}
int main() {
    double *a, *b, *d; CK(hipMallocManaged(&a, 512)); CK(hipMallocManaged(&b, 512)); CK(hipMallocManaged(&d, 512));
    // A one-hot at lane la, B all ones: which D lanes light up -> (block, i) of la; then the converse for B
    printf("A lane -> D lanes (B = 1):\n");
    for (int la = 0; la < 64; ++la) {
        for (int l = 0; l < 64; ++l) { a[l] = l == la ? 1.0 : 0.0; b[l] = 1.0; }
        one<<<1, 64>>>(a, b, d); CK(hipDeviceSynchronize());
        printf("  A%2d:", la); for (int l = 0; l < 64; ++l) if (d[l] != 0.0) printf(" %d", l); printf("\n");
    }
    printf("B lane -> D lanes (A = 1):\n");
    for (int lb = 0; lb < 64; ++lb) {
        for (int l = 0; l < 64; ++l) { b[l] = l == lb ? 1.0 : 0.0; a[l] = 1.0; }
        one<<<1, 64>>>(a, b, d); CK(hipDeviceSynchronize());
        printf("  B%2d:", lb); for (int l = 0; l < 64; ++l) if (d[l] != 0.0) printf(" %d", l); printf("\n");
    }
    // which (A lane, B lane) pairs contribute to D lane 0..: k index pairing
    printf("pairs (A lane, B lane) feeding D lane 0 and D lane 21:\n");
    for (int dl : {0, 21}) {
        printf("  D%2d:", dl);
        for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
            for (int l = 0; l < 64; ++l) { a[l] = l == la ? 1.0 : 0.0; b[l] = l == lb ? 1.0 : 0.0; }
            one<<<1, 64>>>(a, b, d); CK(hipDeviceSynchronize());
            if (d[dl] != 0.0) printf(" (%d,%d)", la, lb);
        }
        printf("\n");
    }
    double *out; CK(hipMalloc(&out, 8 * 1024 * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; ++kind) {
        const int iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (kind == 0) rate<0><<<1024, 256>>>(out, iters); else rate<1><<<1024, 256>>>(out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = 1024.0 * 4 * iters * 4;   // wave-level MFMAs
        const double fl = kind == 0 ? 2048.0 : 512.0;
        printf("%s: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", kind == 0 ? "16x16x4" : "4x4x4 ", ms, n * fl / (ms * 1e-3) / 1e12,
               ms * 1e6 / (n / 1024.0));
    }
    return 0;
}
