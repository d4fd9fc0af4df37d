// Lift-kernel probe: which of {basis loads, MFMAs, stores} bounds the tile loop?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KS, bool LOADS, bool MFMA, bool PIPE, int WROWS>
__global__ __launch_bounds__(256) void lift(const double *Xr, const double *ulift, const double *ref, double *out,
                                            long B, long n_f, long ldo, int tiles_per_wg, int r) {
    constexpr int MT = WROWS / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rowbase = (long)blockIdx.x * (4 * WROWS) + wave * WROWS;
    const int lrow = lane & 15, kgrp = lane >> 4;
    double af[MT][KS];
    for (int mt = 0; mt < MT; ++mt)
        for (int t = 0; t < KS; ++t) { int j = 4 * t + kgrp; af[mt][t] = j < r ? Xr[(rowbase + mt * 16 + lrow) * r + j] : 0.0; }
    const long t0 = (long)blockIdx.y * tiles_per_wg, t1 = min(t0 + tiles_per_wg, n_f / 16);
    double bf[KS], rv = 0.0;
    auto fetch = [&](long it, double (&dst)[KS], double &rd) {
        if (LOADS) {
            const double *uf = ulift + it * KS * 64;
#pragma unroll
            for (int t = 0; t < KS; ++t) dst[t] = uf[t * 64 + lane];
            rd = ref[16 * it + (lane & 15)];
        } else {
#pragma unroll
            for (int t = 0; t < KS; ++t) dst[t] = (double)(it + t);
            rd = 1.0;
        }
    };
    auto tile = [&](long it, const double (&b)[KS], double rr) {
        d4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = d4{0.0, 0.0, 0.0, 0.0};
        if (MFMA) {
#pragma unroll
            for (int t = 0; t < KS; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mt][t], b[t], acc[mt], 0, 0, 0);
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = d4{b[0] + af[mt][0], b[1], b[2], b[3 % KS]};
        }
        double *o = out + (rowbase + kgrp) * ldo + 16 * it + (lane & 15);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) o[(long)(4 * reg + 16 * mt) * ldo] = acc[mt][reg] + rr;
    };
    if (PIPE) {
        if (t0 < t1) fetch(t0, bf, rv);
        for (long it = t0; it < t1; ++it) {
            double bn[KS], rn;
            fetch(it + 1 < t1 ? it + 1 : it, bn, rn);
            tile(it, bf, rv);
#pragma unroll
            for (int t = 0; t < KS; ++t) bf[t] = bn[t];
            rv = rn;
        }
    } else {
        for (long it = t0; it < t1; ++it) { fetch(it, bf, rv); tile(it, bf, rv); }
    }
}


// ALIGNED variant: a wave owns MT tiles whose 16 rows share the 128-byte alignment class of their row start; the
// column window of the tile is shifted so that every 16-lane store segment is one whole 128-byte line.
template <int KS, int MT>
__global__ __launch_bounds__(256) void lift_al(const double *Xr, const double *ut, long ldu, const double *ref, double *out,
                                               long B, long n_f, long ldo, int tiles_per_wg, int r, int c) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wave;
    const long sb = item / c; const int j = (int)(item % c);
    const long R0 = sb * (long)c * 16 * MT + j;                       // rows R0 + c*(16*mt + i)
    const int lrow = lane & 15, kgrp = lane >> 4, col = lane & 15;
    if (R0 + (long)c * (16 * MT - 1) >= B) return;
    const int a = (int)((((unsigned long)out >> 3) + (unsigned long)R0 * ldo) & 15);
    double af[MT][KS];
    for (int mt = 0; mt < MT; ++mt)
        for (int t = 0; t < KS; ++t) { int jj = 4 * t + kgrp; af[mt][t] = jj < r ? Xr[(R0 + (long)c * (16 * mt + lrow)) * r + jj] : 0.0; }
    const long ntl = (n_f + a + 15) / 16;
    const long t0 = (long)blockIdx.y * tiles_per_wg, t1 = min(t0 + tiles_per_wg, ntl);
    double bf[KS], rv = 0.0;
    auto fetch = [&](long it, double (&dst)[KS], double &rd) {
        const double *u = ut + 16 + 16 * it + col - a + (long)kgrp * ldu;
#pragma unroll
        for (int t = 0; t < KS; ++t) dst[t] = u[(long)(4 * t) * ldu];
        long i = 16 * it + col - a; i = i < 0 ? 0 : (i >= n_f ? n_f - 1 : i);
        rd = ref[i];
    };
    auto tile = [&](long it, const double (&b)[KS], double rr, bool guard) {
        d4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mt][t], b[t], acc[mt], 0, 0, 0);
        const long i = 16 * it + col - a;
        double *o = out + (R0 + (long)c * kgrp) * ldo + i;
        if (!guard || (i >= 0 && i < n_f)) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) o[(long)c * (4 * reg + 16 * mt) * ldo] = acc[mt][reg] + rr;
        }
    };
    long it = t0;
    const long tlast = min(t1, (n_f + a) / 16);       // tiles [max(t0,1 if a), tlast) are whole
    if (it < t1 && it == 0 && a) { fetch(it, bf, rv); tile(it, bf, rv, true); ++it; }
    if (it < tlast) {
        fetch(it, bf, rv);
        for (; it < tlast; ++it) {
            double bn[KS], rn;
            fetch(it + 1 < tlast ? it + 1 : it, bn, rn);
            tile(it, bf, rv, false);
#pragma unroll
            for (int t = 0; t < KS; ++t) bf[t] = bn[t];
            rv = rn;
        }
    }
    for (; it < t1; ++it) { fetch(it, bf, rv); tile(it, bf, rv, true); }
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
    printf("%-52s %.3f ms  %.2f TB/s\n", name, ms / it, bytes / (ms / it * 1e-3) / 1e12);
}

int main() {
    const long B = 65536, n_f = 4884; const int r = 30; constexpr int KS = 8;
    double *out, *Xr, *ul, *ref;
    CK(hipMalloc(&out, sizeof(double) * B * 4928)); CK(hipMalloc(&Xr, sizeof(double) * B * 64));
    CK(hipMalloc(&ul, sizeof(double) * 4928 * 64)); CK(hipMalloc(&ref, sizeof(double) * 4928));
    CK(hipMemset(Xr, 0, sizeof(double) * B * 64)); CK(hipMemset(ul, 0, sizeof(double) * 4928 * 64)); CK(hipMemset(ref, 0, sizeof(double) * 4928));
    const double bytes = 8.0 * B * (n_f / 16 * 16);
    const long nt = n_f / 16;
    for (long ldo : {4884L, 4896L}) {
        for (int ys : {2}) {
            const int tpw = (int)((nt + ys - 1) / ys);
            char nm[128];
#define RUN(L, M, P, W) snprintf(nm, 128, "ldo=%ld ys=%d loads=%d mfma=%d pipe=%d rows/wave=%d", ldo, ys, L, M, P, W); \
            timeit(nm, bytes, [&] { lift<KS, L, M, P, W><<<dim3(B / (4 * W), ys), 256>>>(Xr, ul, ref, out, B, n_f, ldo, tpw, r); });
            RUN(false, false, false, 32)
            RUN(true, false, false, 32)
            RUN(true, false, true, 32)
            RUN(false, true, false, 32)
            RUN(true, true, false, 32)
            RUN(true, true, true, 32)
            RUN(true, true, true, 16)
            RUN(true, true, true, 64)
        }
    }

    {
        double *ut; const long ldu = 4928 + 32; CK(hipMalloc(&ut, sizeof(double) * ldu * 64)); CK(hipMemset(ut, 0, sizeof(double) * ldu * 64));
        for (long nf2 : {4884L, 2127L, 4896L}) {
            const long ldo = nf2; const double by = 8.0 * B * nf2;
            int m = (int)(ldo % 16), g = 16; while (m % g) g >>= 1; const int c = 16 / (m ? g : 16);
            for (int ys : {1, 2, 4}) {
                char nm[128];
                const long ntl = (nf2 + 31) / 16; const int tpw = (int)((ntl + ys - 1) / ys);
                snprintf(nm, 128, "ALIGNED n_f=ldo=%ld classes=%d ys=%d MT=2", ldo, c, ys);
                timeit(nm, by, [&] { lift_al<KS, 2><<<dim3(B / 32 / 4, ys), 256>>>(Xr, ut, ldu, ref, out, B, nf2, ldo, tpw, r, c); });
                snprintf(nm, 128, "ALIGNED n_f=ldo=%ld classes=%d ys=%d MT=4", ldo, c, ys);
                timeit(nm, by, [&] { lift_al<KS, 4><<<dim3(B / 64 / 4, ys), 256>>>(Xr, ut, ldu, ref, out, B, nf2, ldo, tpw, r, c); });
                snprintf(nm, 128, "ALIGNED n_f=ldo=%ld classes=%d ys=%d MT=1", ldo, c, ys);
                timeit(nm, by, [&] { lift_al<KS, 1><<<dim3(B / 16 / 4, ys), 256>>>(Xr, ut, ldu, ref, out, B, nf2, ldo, tpw, r, c); });
            }
        }
    }
    return 0;
}
