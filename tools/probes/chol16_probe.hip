// One-wave 16 x 16 Cholesky + inverse: v_readlane broadcasts (qpc::chol16 of rounds 2-5) against DPP row_newbcast broadcasts
// (v_fmac_f64_dpp: the multiplier comes out of the neighbour lane inside the FMA, no SGPR round trip, no hazard s_nop).
// Both forms must give the SAME BITS (same fused operations in the same order); the probe checks that and prints the
// shader clocks per factorisation on one wave of a 512-thread workgroup (the other seven wait at a barrier, as in the kernels).
// Build: hipcc -O3 --offload-arch=gfx950 chol16_probe.hip -o bin/chol16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef double *lptr;
typedef const double *clptr;
struct QPDims { int N, m, po, KT, cond, diagD; };
struct Lds { lptr A, B, Rinv, Ldi, Ls, ks, Qu, ta, tb, tc, part, red; int *flag, *goff; };
namespace wg { typedef double qp_d4 __attribute__((ext_vector_type(4))); }
constexpr int TS = 17, TSZ = 16 * TS;

__device__ __forceinline__ double readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ bool chol16_readlane(lptr T, lptr Rinv) {
    const int lane = threadIdx.x & 63, c = lane & 15, grp = lane >> 4;
    double a[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double t = T[r * TS + c];
        a[r] = grp == 0 ? t : ((grp == 1 && r == c) ? 1.0 : 0.0);
    }
    bool ok = true;
    double piv = readlane_d(a[0], 0);
    ok = ok && (piv > 0.0);
    double di = rsqrt(piv);
    double di2 = di * (1.5 - 0.5 * piv * di * di);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        a[s] *= di2;
        if (s + 1 < 16) {
            a[s + 1] = fma(-readlane_d(a[s], s + 1), a[s], a[s + 1]);
            piv = readlane_d(a[s + 1], s + 1);
            ok = ok && (piv > 0.0);
            di = rsqrt(piv);
            di2 = di * (1.5 - 0.5 * piv * di * di);
        }
#pragma unroll
        for (int r = s + 2; r < 16; ++r) a[r] = fma(-readlane_d(a[s], r), a[s], a[r]);
    }
    if (grp == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r * TS + c] = (r <= c) ? a[r] : 0.0;
    } else if (grp == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Rinv[c * TS + r] = a[r];
    }
    return ok;
}

// acc <- acc - src[lane R of this row of 16 lanes] * oth.  NOP: the DPP operand was written by one of the two VALU instructions before
template <int R, bool NOP>
__device__ __forceinline__ void fnma_bc(double &acc, double src, double oth) {
    if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(oth), "n"(R));
    else asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(oth), "n"(R));
}
template <int R>
__device__ __forceinline__ double mov_bc(double src) {
    double o;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(src), "n"(R));
    return o;
}

template <int S, int R>
struct RowOps {
    static __device__ __forceinline__ void run(double (&a)[16], double (&b)[16]) {
        if constexpr (R < 16) {
            fnma_bc<R, false>(a[R], a[S], a[S]);
            fnma_bc<R, false>(b[R], a[S], b[S]);
            RowOps<S, R + 1>::run(a, b);
        }
    }
};
template <int S>
struct Steps {
    static __device__ __forceinline__ void run(double (&a)[16], double (&b)[16], double di2, bool &ok) {
        if constexpr (S < 16) {
            a[S] *= di2;
            b[S] *= di2;
            double nd = 0.0;
            if constexpr (S + 1 < 16) {
                fnma_bc<S + 1, true>(a[S + 1], a[S], a[S]);
                const double piv = mov_bc<S + 1>(a[S + 1]);
                ok = ok && (piv > 0.0);
                const double di = rsqrt(piv);
                nd = di * (1.5 - 0.5 * piv * di * di);
                fnma_bc<S + 1, false>(b[S + 1], a[S], b[S]);
            }
            RowOps<S, S + 2>::run(a, b);
            Steps<S + 1>::run(a, b, nd, ok);
        }
    }
};

// Every row of 16 lanes carries the whole tile (lane c: column c of the tile in a[], column c of the identity in b[]); rows 1-3 repeat row 0
__device__ __forceinline__ bool chol16_dpp(lptr T, lptr Rinv) {
    const int lane = threadIdx.x & 63, c = lane & 15, grp = lane >> 4;
    double a[16], b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a[r] = T[r * TS + c]; b[r] = r == c ? 1.0 : 0.0; }
    bool ok = true;
    const double piv = mov_bc<0>(a[0]);
    ok = ok && (piv > 0.0);
    const double di = rsqrt(piv);
    Steps<0>::run(a, b, di * (1.5 - 0.5 * piv * di * di), ok);
    if (grp == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r * TS + c] = (r <= c) ? a[r] : 0.0;
    } else if (grp == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Rinv[c * TS + r] = b[r];
    }
    return ok;
}


// ---- hand-ordered form: the pivot chain (fmac -> broadcast -> rsq + third-order correction -> scale) in asm volatile statements, the
// independent row operations of the step placed between its links (in-order issue: what stands between two dependent instructions
// is what hides the latency).  rsq: v_rsq_f64 + the third-order correction of the library's rsqrt (no class test: a pivot that is
// not a positive normal number clears `ok`), WITHOUT the extra Newton step of the readlane form: a rounding-level change.
__device__ __forceinline__ void a_scale(double &a, double y) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(y)); }
template <int R, int NOP>
__device__ __forceinline__ void a_fnma_bc(double &acc, double src, double oth) {
    if constexpr (NOP == 2) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(oth), "n"(R));
    else if constexpr (NOP == 1) asm volatile("s_nop 0\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(oth), "n"(R));
    else asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(oth), "n"(R));
}
template <int R, bool NOP>
__device__ __forceinline__ double a_mov_bc(double src) {
    double o;
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(src), "n"(R));
    else asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(src), "n"(R));
    return o;
}
// independent operation I of step S: I = 0: row S+1 of b; then rows S+2.. of a and b in turn
template <int S, int I>
__device__ __forceinline__ void ind_op(double (&a)[16], double (&b)[16]) {
    constexpr int n = S + 1 < 16 ? 1 + 2 * (14 - S) : 0;
    if constexpr (I < n) {
        if constexpr (I == 0) a_fnma_bc<S + 1, 0>(b[S + 1], a[S], b[S]);
        else {
            constexpr int r = S + 2 + (I - 1) / 2;
            if constexpr ((I - 1) % 2 == 0) a_fnma_bc<r, 0>(a[r], a[S], a[S]);
            else a_fnma_bc<r, 0>(b[r], a[S], b[S]);
        }
    }
}
template <int S, int I0, int I1>
__device__ __forceinline__ void ind_ops(double (&a)[16], double (&b)[16]) {
    if constexpr (I0 < I1) { ind_op<S, I0>(a, b); ind_ops<S, I0 + 1, I1>(a, b); }
}
template <int S>
struct FastSteps {
    static __device__ __forceinline__ void run(double (&a)[16], double (&b)[16], double y, double c375, bool &ok) {
        if constexpr (S < 16) {
            constexpr int n = S + 1 < 16 ? 1 + 2 * (14 - S) : 0;
            a_scale(a[S], y);
            a_scale(b[S], y);
            double yn = 0.0;
            if constexpr (S + 1 < 16) {
                a_fnma_bc<S + 1, 1>(a[S + 1], a[S], a[S]);
                ind_ops<S, 0, 2>(a, b);
                const double p = a_mov_bc<S + 1, (n < 2)>(a[S + 1]);
                ok = ok && (p > 0.0);
                ind_ops<S, 2, 4>(a, b);
                double y0, t, e, g, q;
                asm volatile("v_rsq_f64 %0, %1" : "=v"(y0) : "v"(p));
                ind_ops<S, 4, 9>(a, b);
                if constexpr (n < 5) asm volatile("s_nop 0\n\tv_mul_f64 %0, %1, %2" : "=v"(t) : "v"(p), "v"(y0));
                else asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(p), "v"(y0));
                ind_ops<S, 9, 11>(a, b);
                asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(t), "v"(y0));
                ind_ops<S, 11, 13>(a, b);
                asm volatile("v_fma_f64 %0, %1, %2, 0.5" : "=v"(g) : "v"(e), "v"(c375));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(q) : "v"(y0), "v"(e));
                ind_ops<S, 13, 15>(a, b);
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(yn) : "v"(q), "v"(g), "v"(y0));
                ind_ops<S, 15, 64>(a, b);
            }
            FastSteps<S + 1>::run(a, b, yn, c375, ok);
        }
    }
};
__device__ __forceinline__ bool chol16_fast(lptr T, lptr Rinv) {
    const int lane = threadIdx.x & 63, c = lane & 15, grp = lane >> 4;
    double a[16], b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a[r] = T[r * TS + c]; b[r] = r == c ? 1.0 : 0.0; }
    bool ok = true;
    const double c375 = 0.375;
    const double p = a_mov_bc<0, true>(a[0]);
    ok = ok && (p > 0.0);
    double y0, t, e, g, q, y;
    asm volatile("v_rsq_f64 %0, %1" : "=v"(y0) : "v"(p));
    asm volatile("s_nop 0\n\tv_mul_f64 %0, %1, %2" : "=v"(t) : "v"(p), "v"(y0));
    asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(t), "v"(y0));
    asm volatile("v_fma_f64 %0, %1, %2, 0.5" : "=v"(g) : "v"(e), "v"(c375));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(q) : "v"(y0), "v"(e));
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(y) : "v"(q), "v"(g), "v"(y0));
    FastSteps<0>::run(a, b, y, c375, ok);
    if (grp == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r * TS + c] = (r <= c) ? a[r] : 0.0;
    } else if (grp == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Rinv[c * TS + r] = b[r];
    }
    return ok;
}

// ---- the product's form (csrc/locp_cond.h: qpc::chol16), copied verbatim: no s_nop in the asm -- build through tools/hipcc_guarded.sh
namespace prod {
// acc <- acc - src[lane R of this row of 16 lanes] * oth
template <int R>
__device__ __forceinline__ void p_fnma_bc(double &acc, double src, double oth) {
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(oth), "n"(R));
}
template <int R>
__device__ __forceinline__ double p_mov_bc(double src) {
    double o;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(src), "n"(R));
    return o;
}
// 1 / sqrt(p) for a positive normal p: v_rsq_f64 and the third-order correction of the library's rsqrt (without its class test:
// the caller flags a pivot that is not positive)
__device__ __forceinline__ double rsq3(double p) {
    const double y0 = __builtin_amdgcn_rsq(p), e = fma(-(p * y0), y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
template <int S, int R, int NPC>
__device__ __forceinline__ void p_rows(double (&a)[16], double (&b)[16]) {
    if constexpr (R < NPC) {
        p_fnma_bc<R>(a[R], a[S], a[S]);
        p_fnma_bc<R>(b[R], a[S], b[S]);
        p_rows<S, R + 1, NPC>(a, b);
    }
}
// A pivot that is not positive shows in the LAST scale factor: rsq of a negative number is NaN, of zero infinite, and either
// reaches every later pivot through the row operations -- one test at the end instead of one per pivot.
template <int S, int NPC>
__device__ __forceinline__ void p_steps(double (&a)[16], double (&b)[16], double y, double &ylast) {
    if constexpr (S < NPC) {
        a[S] *= y;
        b[S] *= y;
        double yn = 0.0;
        if constexpr (S + 1 < NPC) {
            p_fnma_bc<S + 1>(a[S + 1], a[S], a[S]);
            yn = rsq3(p_mov_bc<S + 1>(a[S + 1]));
            p_fnma_bc<S + 1>(b[S + 1], a[S], b[S]);
        } else ylast = y;
        p_rows<S, S + 2, NPC>(a, b);
        p_steps<S + 1, NPC>(a, b, yn, ylast);
    }
}
// STORE_T: write the factor back (qpc::r_times / rT_times read the diagonal tiles; the lean kernels only ever use Rinv).
// NPC: only the leading NPC x NPC block of the tile is not the identity (short horizons, ql::ipm_wave: N p_o < 16 outputs): the
// pivots and row operations beyond it are no-ops and are left out -- 45 row operations instead of 120 at NPC = 10.
template <bool STORE_T = true, int NPC = 16>
__device__ __forceinline__ bool chol16_product(lptr T, lptr Rinv) {
    static_assert(NPC >= 1 && NPC <= 16, "p: 1 <= NPC <= 16");
    const int lane = threadIdx.x & 63, c = lane & 15, grp = lane >> 4;
    double a[16], b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a[r] = T[r * TS + c]; b[r] = r == c ? 1.0 : 0.0; }
    double ylast = 0.0;
    p_steps<0, NPC>(a, b, rsq3(p_mov_bc<0>(a[0])), ylast);
    const bool ok = ylast > 0.0 && ylast < INFINITY;
    if (grp == 0) {
        if constexpr (STORE_T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r * TS + c] = (r <= c) ? a[r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Rinv[c * TS + r] = b[r];          // b[r] = M[r][c] = Rinv[c][r]
    }
    return ok;
}

}

// readlane form with the product's rsq3 (bit-for-bit reference of prod::chol16_product)
__device__ __forceinline__ bool chol16_readlane3(lptr T, lptr Rinv) {
    const int lane = threadIdx.x & 63, c = lane & 15, grp = lane >> 4;
    double a[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const double t = T[r * TS + c]; a[r] = grp == 0 ? t : ((grp == 1 && r == c) ? 1.0 : 0.0); }
    bool ok = true;
    double piv = readlane_d(a[0], 0);
    ok = ok && (piv > 0.0);
    double di2 = prod::rsq3(piv);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        a[s] *= di2;
        if (s + 1 < 16) {
            a[s + 1] = fma(-readlane_d(a[s], s + 1), a[s], a[s + 1]);
            piv = readlane_d(a[s + 1], s + 1);
            ok = ok && (piv > 0.0);
            di2 = prod::rsq3(piv);
        }
#pragma unroll
        for (int r = s + 2; r < 16; ++r) a[r] = fma(-readlane_d(a[s], r), a[s], a[r]);
    }
    if (grp == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r * TS + c] = (r <= c) ? a[r] : 0.0;
    } else if (grp == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Rinv[c * TS + r] = a[r];
    }
    return ok;
}

template <int WHICH>
__global__ __launch_bounds__(512) void probe(const double *A, double *outT, double *outR, long long *t, int reps) {
    __shared__ double sm[4 * TSZ];
    lptr T0 = sm, T = sm + TSZ, R = sm + 2 * TSZ;
    const int tid = threadIdx.x;
    for (int i = tid; i < 256; i += 512) T0[(i >> 4) * TS + (i & 15)] = A[i];
    __syncthreads();
    long long c0 = 0, c1 = 0;
    int ok = 1;
    if (tid < 64) {
        c0 = clock64();
        for (int it = 0; it < reps; ++it) {
            for (int i = tid; i < 256; i += 64) T[(i >> 4) * TS + (i & 15)] = T0[(i >> 4) * TS + (i & 15)];
            __builtin_amdgcn_wave_barrier();
            const bool o = WHICH == 0 ? chol16_readlane(T, R) : (WHICH == 1 ? chol16_dpp(T, R) : (WHICH == 2 ? chol16_fast(T, R) : (WHICH == 3 ? prod::chol16_product<true, 16>(T, R) : chol16_readlane3(T, R))));
            ok = ok && o;
            __builtin_amdgcn_wave_barrier();
        }
        c1 = clock64();
    }
    __syncthreads();
    for (int i = tid; i < 256; i += 512) { outT[i] = T[(i >> 4) * TS + (i & 15)]; outR[i] = R[(i >> 4) * TS + (i & 15)]; }
    if (tid == 0) { t[0] = (c1 - c0) / reps; t[1] = ok; }
}

int main() {
    std::vector<double> G(16 * 24), A(256, 0.0);
    srand(3);
    for (auto &g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = i == j ? 1.0 : 0.0; for (int k = 0; k < 24; ++k) s += G[i * 24 + k] * G[j * 24 + k]; A[i * 16 + j] = s; }
    double *dA, *dT, *dR; long long *dt;
    hipMalloc(&dA, 256 * 8); hipMalloc(&dT, 2 * 256 * 8); hipMalloc(&dR, 2 * 256 * 8); hipMalloc(&dt, 4 * 8);
    double *dT2, *dR2; hipMalloc(&dT2, 256 * 8); hipMalloc(&dR2, 256 * 8);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    std::vector<double> hT(512), hR(512); long long ht[4];
    for (int pass = 0; pass < 2; ++pass) {
        probe<0><<<1, 512>>>(dA, dT, dR, dt, 200); hipDeviceSynchronize();
        hipMemcpy(ht, dt, 16, hipMemcpyDeviceToHost);
        probe<1><<<1, 512>>>(dA, dT + 256, dR + 256, dt + 2, 200); hipDeviceSynchronize();
        hipMemcpy(ht + 2, dt + 2, 16, hipMemcpyDeviceToHost);
        long long hf[2];
        probe<2><<<1, 512>>>(dA, dT2, dR2, dt + 2, 200); hipDeviceSynchronize();
        hipMemcpy(hf, dt + 2, 16, hipMemcpyDeviceToHost);
        printf("pass %d: readlane %lld clocks (ok %lld), dpp %lld clocks (ok %lld), hand-ordered %lld clocks (ok %lld)\n", pass, ht[0], ht[1], ht[2], ht[3], hf[0], hf[1]);
    }
    hipMemcpy(hT.data(), dT, 512 * 8, hipMemcpyDeviceToHost); hipMemcpy(hR.data(), dR, 512 * 8, hipMemcpyDeviceToHost);
    const int sameT = memcmp(hT.data(), hT.data() + 256, 256 * 8) == 0, sameR = memcmp(hR.data(), hR.data() + 256, 256 * 8) == 0;
    double err = 0.0;                                  // R^T R against A
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0.0; for (int k = 0; k < 16; ++k) s += hT[256 + k * 16 + i] * hT[256 + k * 16 + j]; err = fmax(err, fabs(s - A[i * 16 + j])); }
    printf("factor bits equal: %d, inverse bits equal: %d, max |R^T R - A| (dpp) %.3e\n", sameT, sameR, err);
    std::vector<double> fT(256), fR(256);
    hipMemcpy(fT.data(), dT2, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(fR.data(), dR2, 256 * 8, hipMemcpyDeviceToHost);
    double dT_ = 0.0, dR_ = 0.0, errf = 0.0, inv = 0.0;
    for (int i = 0; i < 256; ++i) { dT_ = fmax(dT_, fabs(fT[i] - hT[i])); dR_ = fmax(dR_, fabs(fR[i] - hR[i])); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0.0, w = 0.0;
        for (int k = 0; k < 16; ++k) { s += fT[k * 16 + i] * fT[k * 16 + j]; w += fT[i * 16 + k] * fR[k * 16 + j]; }
        errf = fmax(errf, fabs(s - A[i * 16 + j])); inv = fmax(inv, fabs(w - (i == j ? 1.0 : 0.0)));
    }
    {   // product form against the readlane form with the same rsq3: bits
        double *dT3, *dR3; hipMalloc(&dT3, 512 * 8); hipMalloc(&dR3, 512 * 8);
        probe<3><<<1, 512>>>(dA, dT3, dR3, dt, 200); hipDeviceSynchronize();
        long long h3[2]; hipMemcpy(h3, dt, 16, hipMemcpyDeviceToHost);
        probe<4><<<1, 512>>>(dA, dT3 + 256, dR3 + 256, dt, 200); hipDeviceSynchronize();
        std::vector<double> pT(512), pR(512);
        hipMemcpy(pT.data(), dT3, 512 * 8, hipMemcpyDeviceToHost); hipMemcpy(pR.data(), dR3, 512 * 8, hipMemcpyDeviceToHost);
        const int eT = memcmp(pT.data(), pT.data() + 256, 256 * 8) == 0, eR = memcmp(pR.data(), pR.data() + 256, 256 * 8) == 0;
        printf("product form: %lld clocks (ok %lld); bits equal to the readlane form with rsq3: factor %d inverse %d\n", h3[0], h3[1], eT, eR);
        if (!(eT && eR)) return 2;
    }
    printf("hand-ordered against readlane: max |dR| %.3e, max |dRinv| %.3e; |R^T R - A| %.3e, |R Rinv - I| %.3e\n", dT_, dR_, errf, inv);
    return !(sameT && sameR && errf < 1e-13 && inv < 1e-13);
}
