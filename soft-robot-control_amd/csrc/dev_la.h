// Workgroup-cooperative small dense linear algebra in LDS / L2 (one workgroup = one problem).
// Conventions: every routine is called by ALL threads of the workgroup; inputs must be visible
// (caller has synchronised); routines that write shared data end with __syncthreads().
#pragma once
#include <hip/hip_runtime.h>

// The thread index as the kernels' inlined phases read it: through an asm statement the optimiser cannot look into, so that nothing
// derived from it (the LDS / global addresses of "my" elements in every phase) is loop invariant in its eyes.  Hoisted out of the
// interior-point loop those values -- hundreds of them in a kernel that is one inlined body -- were spilled and reloaded at every phase
// boundary (1040 B of scratch per lane in the C2 kernel, reloads that miss L2 under the 4096-rollout launch); recomputing them is one or
// two VALU instructions each: 364 B of scratch, C2 45.7 -> 39.5 ms per 4096 rollouts, one rollout 0.92 -> 0.83 ms per SCP iteration
// (round 5; -DSRH_PLAIN_TID: the plain register read, for A/B).  Same arithmetic, same results.
#ifndef SRH_PLAIN_TID
__device__ __forceinline__ unsigned srh_tid_now() { unsigned t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#define SRH_TID srh_tid_now()
#else
#define SRH_TID threadIdx.x
#endif

// Address-space qualified pointers.  Generic pointers make hipcc emit FLAT loads/stores (slow path for
// LDS, and they tie up both memory counters); typing LDS and global buffers explicitly gives ds_* and
// global_* instructions even across non-inlined calls.
typedef __attribute__((address_space(3))) double ld_t;
typedef __attribute__((address_space(1))) double gd_t;
typedef __attribute__((address_space(3))) int li_t;
typedef __attribute__((address_space(1))) int gi_t;
using lptr = ld_t *;
using clptr = const ld_t *;
using gptr = gd_t *;
using cgptr = const gd_t *;
using giptr = gi_t *;
using cgiptr = const gi_t *;
using liptr = li_t *;

namespace wg {

__device__ __forceinline__ int tid() { return SRH_TID; }
__device__ __forceinline__ int nthr() { return blockDim.x; }

// ---- reductions over the workgroup (scratch: >= 16 doubles of LDS) ---------------------------
// ---- sums over aligned groups of G = 4 / 8 / 16 lanes by DPP moves (no LDS crossbar: ~3 VALU instructions per step
// instead of two ds_bpermute round trips); every lane of the group receives the sum.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi2, lo2);
}
// the value of lane ^ 16 (the neighbouring row of 16 lanes).  v_permlane16_swap_b32 (gfx950) swaps the odd rows of its first operand
// with the even rows of its second: with the same value in both, the first result carries the even rows' values in both rows of a
// pair and the second the odd rows' -- two VALU instructions and a select per 32 bits instead of a ds_bpermute round trip.
__device__ __forceinline__ double xor16(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const bool odd = (SRH_TID >> 4) & 1;
    return __hiloint2double(odd ? b[0] : b[1], odd ? a[0] : a[1]);
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    static_assert(G == 4 || G == 8 || G == 16, "group_sum: G must be 4, 8 or 16");
    v += dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                       // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp_mov<0x141>(v);    // row_half_mirror: lane i <- lane 7 - i of its 8
    if constexpr (G >= 16) v += dpp_mov<0x140>(v);   // row_mirror:      lane i <- lane 15 - i of its 16
    return v;
}

// ---- whole-wave reductions by DPP: four steps inside the rows of 16, row_bcast15 / row_bcast31 across the rows, the
// total of lane 63 broadcast with v_readlane (fixed tree: deterministic; every lane receives the same value)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov_rows(double v, double fill) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int flo = __double2loint(fill), fhi = __double2hiint(fill);
    const int lo2 = __builtin_amdgcn_update_dpp(flo, lo, CTRL, ROWMASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(fhi, hi, CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi2, lo2);
}
template <int OP>     // 0 sum, 1 max, 2 min
__device__ __forceinline__ double wave_reduce(double v) {
    auto f = [](double a, double b) { return OP == 0 ? a + b : (OP == 1 ? fmax(a, b) : fmin(a, b)); };
    const double id = OP == 0 ? 0.0 : (OP == 1 ? -INFINITY : INFINITY);
    v = f(v, dpp_mov<0xB1>(v));
    v = f(v, dpp_mov<0x4E>(v));
    v = f(v, dpp_mov<0x141>(v));
    v = f(v, dpp_mov<0x140>(v));
    v = f(v, dpp_mov_rows<0x142, 0xa>(v, id));       // row_bcast15 into rows 1 and 3
    v = f(v, dpp_mov_rows<0x143, 0xc>(v, id));       // row_bcast31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<0>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<1>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<2>(v); }
// op: 0 sum, 1 max, 2 min.  Result broadcast to every thread.  Deterministic (fixed tree).
__device__ inline double reduce(double v, int op, lptr scratch) {
    double w = op == 0 ? wave_sum(v) : (op == 1 ? wave_max(v) : wave_min(v));
    const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), nw = (blockDim.x + 63) >> 6;
    __syncthreads();  // scratch may still be read from a previous reduction
    if ((SRH_TID & 63) == 0) scratch[wave] = w;
    __syncthreads();
    double r = scratch[0];
    for (int i = 1; i < nw; ++i) r = op == 0 ? r + scratch[i] : (op == 1 ? fmax(r, scratch[i]) : fmin(r, scratch[i]));
    return r;
}

// y (len) = sum over i<rows of M[i][j] * v[i]   (i.e. y = M^T v for row-major M (rows x len)), M in
// global/L2 (coalesced along j) or LDS; v, y in LDS.  Adds `add` if non-null.  part: (nthr) doubles.
template <typename MP, typename AP>
__device__ inline void matTvec(lptr y, MP M, int ldm, int rows, int len, clptr v, AP add, lptr part) {
    // threads = (slice, j): slice s handles rows s, s+S, ...
    const int S = max(1, (int)blockDim.x / len);
    const int j = SRH_TID % len, s = SRH_TID / len;
    double acc = 0.0;
    if (s < S) {
#pragma unroll 4
        for (int i = s; i < rows; i += S) acc = fma(M[i * ldm + j], v[i], acc);
        part[s * len + j] = acc;
    }
    __syncthreads();
    if (SRH_TID < len) {
        double r = add ? add[SRH_TID] : 0.0;
        for (int q = 0; q < S; ++q) r += part[q * len + SRH_TID];
        y[SRH_TID] = r;
    }
    __syncthreads();
}

// Cholesky factor L (lower, row-major m x m, m <= 16) of a tiny SPD matrix by thread 0.  Returns false
// (to all threads, via flag in LDS) if not positive definite.  With allow_shift a breakdown caused by
// round-off in a nearly singular matrix is retried with a growing diagonal shift (inexact Newton step;
// the interior-point iteration corrects it).
__device__ inline bool chol_factor(clptr Q, lptr Lbuf, int m, liptr flag, bool allow_shift = false) {
    if (SRH_TID == 0) {
        bool ok = false;
        double dmax = 0.0;
        for (int i = 0; i < m; ++i) dmax = fmax(dmax, fabs(Q[i * m + i]));
        double shift = 0.0;
        for (int attempt = 0; attempt < (allow_shift ? 8 : 1) && !ok; ++attempt) {
            ok = true;
            for (int i = 0; i < m && ok; ++i) {
                for (int j = 0; j <= i; ++j) {
                    double sum = Q[i * m + j] + (i == j ? shift : 0.0);
                    for (int k = 0; k < j; ++k) sum -= Lbuf[i * m + k] * Lbuf[j * m + k];
                    if (i == j) {
                        if (!(sum > 0.0)) { ok = false; break; }
                        Lbuf[i * m + i] = sqrt(sum);
                    } else {
                        Lbuf[i * m + j] = sum / Lbuf[j * m + j];
                    }
                }
            }
            shift = (shift == 0.0) ? 1e-14 * dmax : shift * 100.0;
        }
        *flag = ok ? 1 : 0;
    }
    __syncthreads();
    return *flag != 0;
}

// x = -(L L^T)^-1 b for one right-hand side held by the calling thread: b, x strided arrays (m <= 16)
template <typename LP, typename BP, typename XP>
__device__ __forceinline__ void chol_solve_neg(LP L, int m, BP b, int bstride, XP x, int xstride) {
    double y[16];
    for (int i = 0; i < m; ++i) {
        double sum = b[i * bstride];
        for (int k = 0; k < i; ++k) sum -= L[i * m + k] * y[k];
        y[i] = sum / L[i * m + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double sum = y[i];
        for (int k = i + 1; k < m; ++k) sum -= L[k * m + i] * y[k];
        y[i] = sum / L[i * m + i];
    }
    for (int i = 0; i < m; ++i) x[i * xstride] = -y[i];
}


// ---- register-resident Cholesky of a tiny SPD matrix (compile-time size) ------------------------
// L (lower, row-major M x M) and the reciprocals of its diagonal; returns false if not positive
// definite.  `shift` is added to the diagonal.
template <int M, typename QP>
__device__ __forceinline__ bool chol_reg(QP Q, int ldq, double shift, double (&Lr)[M * M], double (&inv)[M]) {
    double a[M * M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[i * M + j] = Q[i * ldq + j] + (i == j ? shift : 0.0);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double sum = a[i * M + j];
#pragma unroll
            for (int k = 0; k < j; ++k) sum -= Lr[i * M + k] * Lr[j * M + k];
            if (i == j) {
                if (!(sum > 0.0)) ok = false;
                const double ri = rsqrt(sum);              // one reciprocal square root instead of sqrt + divide
                inv[i] = ri;
                Lr[i * M + i] = sum * ri;
            } else {
                Lr[i * M + j] = sum * inv[j];
            }
        }
    }
    return ok;
}

// x = -(L L^T)^-1 b with L in registers (inv = 1/diag(L))
template <int M, typename BP, typename XP>
__device__ __forceinline__ void chol_solve_neg_reg(const double (&Lr)[M * M], const double (&inv)[M], BP b,
                                                   int bstride, XP x, int xstride) {
    double y[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double sum = b[i * bstride];
#pragma unroll
        for (int k = 0; k < i; ++k) sum -= Lr[i * M + k] * y[k];
        y[i] = sum * inv[i];
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        double sum = y[i];
#pragma unroll
        for (int k = i + 1; k < M; ++k) sum -= Lr[k * M + i] * y[k];
        y[i] = sum * inv[i];
    }
#pragma unroll
    for (int i = 0; i < M; ++i) x[i * xstride] = -y[i];
}

// ------------------------------------------------------------------ f64 MFMA product on LDS operands
typedef double qp_d4 __attribute__((ext_vector_type(4)));

// C[i][j] = sum_{k<K} Lm[k][i] * Rm[k][j]  for i < 16*MT, j < 16*NTl.  Lm, Rm: k-major rows (K x ld) in
// LDS, K a multiple of 4 (zero padded).  Rows i >= vrows of C are stored as exact zeros; rows >= srows are
// not stored at all.
// v_mfma_f64_16x16x4: A lane l holds Lm^T[i=l&15][k=l>>4], B lane holds Rm[k=l>>4][j=l&15];
// D reg q of lane l is C[row = (l>>4) + 4q][col = l&15].
__device__ __forceinline__ void mfma_atb(lptr C, int ldc, clptr Lm, clptr Rm, int K, int MT, int NTl, int ld, int vrows,
                                         int srows = 1 << 30) {
    const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), lane = SRH_TID & 63, nw = blockDim.x >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    const int ntiles = MT * NTl;
    for (int t0 = wave; t0 < ntiles; t0 += 2 * nw) {
        const int t1 = t0 + nw;
        const bool has1 = t1 < ntiles;
        const int ti0 = t0 / NTl, tj0 = t0 - ti0 * NTl;
        const int ti1 = has1 ? t1 / NTl : ti0, tj1 = has1 ? t1 - ti1 * NTl : tj0;
        clptr la0 = Lm + kk * ld + 16 * ti0 + l16, rb0 = Rm + kk * ld + 16 * tj0 + l16;
        clptr la1 = Lm + kk * ld + 16 * ti1 + l16, rb1 = Rm + kk * ld + 16 * tj1 + l16;
        qp_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        int k0 = 0;
        // 4 k-steps per trip: 16 independent LDS reads in flight before the 8 MFMAs consume them
        for (; k0 + 16 <= K; k0 += 16) {
            double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o = (k0 + 4 * u) * ld;
                a0[u] = la0[o]; b0[u] = rb0[o]; a1[u] = la1[o]; b1[u] = rb1[o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc1, 0, 0, 0);
            }
        }
        for (; k0 < K; k0 += 4) {
            const double a0 = la0[k0 * ld], b0 = rb0[k0 * ld];
            const double a1 = la1[k0 * ld], b1 = rb1[k0 * ld];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r0 = 16 * ti0 + kk + 4 * q;
            if (r0 < srows) C[r0 * ldc + 16 * tj0 + l16] = r0 < vrows ? acc0[q] : 0.0;
            if (has1) {
                const int r1 = 16 * ti1 + kk + 4 * q;
                if (r1 < srows) C[r1 * ldc + 16 * tj1 + l16] = r1 < vrows ? acc1[q] : 0.0;
            }
        }
    }
    __syncthreads();
}


}  // namespace wg
