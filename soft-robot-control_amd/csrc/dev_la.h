// Workgroup-cooperative small dense linear algebra in LDS / L2 (one workgroup = one problem).
// Conventions: every routine is called by ALL threads of the workgroup; inputs must be visible
// (caller has synchronised); routines that write shared data end with __syncthreads().
#pragma once
#include <hip/hip_runtime.h>

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
__device__ inline double reduce(double v, int op, double *scratch) {
    double w = op == 0 ? wave_sum(v) : (op == 1 ? wave_max(v) : wave_min(v));
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();  // scratch may still be read from a previous reduction
    if ((threadIdx.x & 63) == 0) scratch[wave] = w;
    __syncthreads();
    double r = scratch[0];
    for (int i = 1; i < nw; ++i) r = op == 0 ? r + scratch[i] : (op == 1 ? fmax(r, scratch[i]) : fmin(r, scratch[i]));
    return r;
}

// ---- GEMM on LDS operands: C (M x N) = op(A) (M x K) * B (K x N) -----------------------------
// Row-major, leading dimensions in doubles; ldb/ldc multiples of 4 and 32-byte aligned bases so the
// 1x4 register block can use 16-byte LDS accesses; columns N..roundup4(N) of B must be finite.
template <bool TRANS_A>
__device__ inline void gemm(double *__restrict__ C, int ldc, const double *__restrict__ A, int lda,
                            const double *__restrict__ B, int ldb, int M, int N, int K) {
    const int nq = (N + 3) >> 2;
    const int items = M * nq;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int i = it / nq, j0 = (it - i * nq) << 2;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
        const double *bp = B + j0;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const double a = TRANS_A ? A[k * lda + i] : A[i * lda + k];
            const double2 b01 = *reinterpret_cast<const double2 *>(bp + k * ldb);
            const double2 b23 = *reinterpret_cast<const double2 *>(bp + k * ldb + 2);
            c0 = fma(a, b01.x, c0);
            c1 = fma(a, b01.y, c1);
            c2 = fma(a, b23.x, c2);
            c3 = fma(a, b23.y, c3);
        }
        double *cp = C + i * ldc + j0;
        *reinterpret_cast<double2 *>(cp) = double2{c0, c1};
        *reinterpret_cast<double2 *>(cp + 2) = double2{c2, c3};
    }
    __syncthreads();
}

// y (len) = sum over i<rows of M[i][j] * v[i]   (i.e. y = M^T v for row-major M (rows x len)), M in
// global/L2 (coalesced along j) or LDS; v, y in LDS.  Adds `add` if non-null.  part: (nthr) doubles.
__device__ inline void matTvec(double *y, const double *__restrict__ M, int ldm, int rows, int len,
                               const double *v, const double *add, double *part) {
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
__device__ inline bool chol_factor(const double *Q, double *Lbuf, int m, int *flag, bool allow_shift = false) {
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
__device__ __forceinline__ void chol_solve_neg(const double *L, int m, const double *b, int bstride, double *x,
                                               int xstride) {
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

}  // namespace wg
