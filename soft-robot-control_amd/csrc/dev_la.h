// Workgroup-cooperative small dense linear algebra in LDS / L2 (one workgroup = one problem).
// Conventions: every routine is called by ALL threads of the workgroup; inputs must be visible
// (caller has synchronised); routines that write shared data end with __syncthreads().
#pragma once
#include <hip/hip_runtime.h>

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

__device__ __forceinline__ int tid() { return threadIdx.x; }
__device__ __forceinline__ int nthr() { return blockDim.x; }

// ---- reductions over the workgroup (scratch: >= 16 doubles of LDS) ---------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
// op: 0 sum, 1 max, 2 min.  Result broadcast to every thread.  Deterministic (fixed tree).
__device__ inline double reduce(double v, int op, lptr scratch) {
    double w = op == 0 ? wave_sum(v) : (op == 1 ? wave_max(v) : wave_min(v));
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();  // scratch may still be read from a previous reduction
    if ((threadIdx.x & 63) == 0) scratch[wave] = w;
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
    const int j = threadIdx.x % len, s = threadIdx.x / len;
    double acc = 0.0;
    if (s < S) {
#pragma unroll 4
        for (int i = s; i < rows; i += S) acc = fma(M[i * ldm + j], v[i], acc);
        part[s * len + j] = acc;
    }
    __syncthreads();
    if (threadIdx.x < len) {
        double r = add ? add[threadIdx.x] : 0.0;
        for (int q = 0; q < S; ++q) r += part[q * len + threadIdx.x];
        y[threadIdx.x] = r;
    }
    __syncthreads();
}

// Cholesky factor L (lower, row-major m x m, m <= 16) of a tiny SPD matrix by thread 0.  Returns false
// (to all threads, via flag in LDS) if not positive definite.  With allow_shift a breakdown caused by
// round-off in a nearly singular matrix is retried with a growing diagonal shift (inexact Newton step;
// the interior-point iteration corrects it).
__device__ inline bool chol_factor(clptr Q, lptr Lbuf, int m, liptr flag, bool allow_shift = false) {
    if (threadIdx.x == 0) {
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
                const double dd = sqrt(sum);
                Lr[i * M + i] = dd;
                inv[i] = 1.0 / dd;
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

}  // namespace wg
